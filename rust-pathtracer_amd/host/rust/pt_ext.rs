//! Optional bindings for the entry points libpt_hip.so adds to the reference's C ABI (include/pt_api.h,
//! part 2).  NOT compiled in the environment this backend was built in (no cargo there).  The reference's own
//! extern blocks (src/main.rs:9-32, src/guiding.rs:31-68) need no change; add `mod pt_ext;` to use these.
#![allow(dead_code)]
use std::os::raw::{c_char, c_float, c_int, c_void};

pub const PT_MODE_STRICT: c_int = 0;
pub const PT_MODE_CONVERGE: c_int = 1;
pub const PT_PIPELINE_AUTO: c_int = -1;
pub const PT_PIPELINE_STAGED: c_int = 0;
pub const PT_PIPELINE_FUSED: c_int = 1;

/// `int (*)(void* user)`: add the caller's statistics buffer up over all ranks, in place.
pub type PtReduceFn = Option<unsafe extern "C" fn(user: *mut c_void) -> c_int>;

extern "C" {
    pub fn pt_set_mode(mode: c_int);
    pub fn pt_set_pipeline(pipeline: c_int);
    pub fn pt_set_device(device: c_int);
    pub fn pt_set_frame(frame: c_int);
    pub fn pt_set_adaptive_round(samples: c_int);
    pub fn pt_get_ray_counts(radiance: *mut u64, shadow: *mut u64);
    pub fn pt_last_render_ms() -> f64;

    pub fn pt_scene_cornell();
    pub fn pt_scene_triangles(vertices: *const c_float, n_vertices: c_int, indices: *const u32,
                              n_triangles: c_int, materials11: *const c_float);
    pub fn pt_scene_load_obj(path: *const c_char) -> c_int;
    pub fn pt_scene_set_light(pos: *const c_float, emit: *const c_float, normal: *const c_float,
                              half: *const c_float);

    pub fn pt_demosaic_f32(raw: *const c_float, w: c_int, h: c_int, pattern: c_int, algo: c_int,
                           out_rgb: *mut c_float) -> c_int;
    pub fn pt_write_exr(path: *const c_char, w: c_int, h: c_int, rgb: *const c_float, exposure: c_float) -> c_int;

    // one process per GPU: device pointers, sums to all-reduce
    pub fn pt_render_accumulate_device(w: c_int, h: c_int, sample_begin: c_int, sample_count: c_int,
                                       max_depth: c_int, frame: c_int, d_sum_rgb: *mut c_float) -> c_int;
    pub fn pt_render_adaptive_shared_device(w: c_int, h: c_int, sample_begin: c_int, sample_count: c_int,
                                            total_spp: c_int, max_depth: c_int, frame: c_int,
                                            noise_threshold: c_float, min_adaptive_samples: c_int,
                                            d_stats: *mut c_float, reduce: PtReduceFn, user: *mut c_void) -> c_int;
    pub fn pt_pgl_samples_add_batch(samples: *mut c_void, records: *const c_void, count: u32);
}

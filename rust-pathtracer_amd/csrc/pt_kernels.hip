// pt_kernels.hip -- wavefront path-tracing kernels for gfx950 (MI355X).
//
// Work unit: a path slot = (pixel, sample chunk).  A slot runs its samples one after the other
// (the reference's raygen does the same inside one thread, optix_device.cu:953-1053), so a
// chunk's PCG stream is consumed in the reference's order.  With one chunk per pixel a slot is
// a pixel; converge-mode renders may split a pixel's sample range into a few chunks that run
// side by side (slot = chunk * npx + pixel), which keeps the GPU full while the pixels with the
// longest paths finish.  Queues hold the ids of slots that still have samples to take; they
// are compacted with wave64 ballot + popcount and one atomic per wave.
//
// Staged pipeline (one iteration = one bounce of every live path):
//   generate : fold finished samples into the pixel sums, start the next sample, compact
//   extend   : closest-hit BVH traversal of the radiance rays            -> hit records
//   shade    : BSDF / NEE / guided sampling / Russian roulette            -> shadow queue
//   occlude  : first-hit traversal of the shadow rays, add unoccluded NEE -> path sums
// Fused pipeline: the same device functions chained per lane with the path in registers
// for several bounces per launch (no hit or shadow records in HBM).
#include "pt_kernels.h"

#include <cstdio>
#include <cstdlib>

#include "pt_bvh.h"
#include "pt_shade.h"

namespace pt {

#define PT_KCHECK()                                                                      \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "HIP launch error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      abort();                                                                           \
    }                                                                                    \
  } while (0)

// 16-byte aligned: the fused kernels keep 16-byte words behind the stacks (the staged scene).  LDS declared
// statically inside a kernel comes before this array and must keep that alignment too: four stray bytes there
// made the fused kernel 2.3 times slower.
extern __shared__ __attribute__((aligned(16))) StackEntry g_lds_stack[];

#ifdef PT_PROFILE
__device__ unsigned long long g_prof_global[64];
__device__ __forceinline__ void prof_begin() {
  if (threadIdx.x < 64) g_prof_lds[threadIdx.x] = 0u;
  __syncthreads();
}
__device__ __forceinline__ void prof_end() {
  __syncthreads();
  if (threadIdx.x < 64 && g_prof_lds[threadIdx.x]) atomicAdd(&g_prof_global[threadIdx.x], (unsigned long long)g_prof_lds[threadIdx.x]);
}
#else
__device__ __forceinline__ void prof_begin() {}
__device__ __forceinline__ void prof_end() {}
#endif

// Per-lane traversal stack in LDS, entry (level, lane) at [level * blockDim + lane]:
// consecutive lanes touch consecutive 8-byte words, so pushes and pops are conflict-free.
struct LdsStack {
  StackEntry* base;
  int stride;
  __device__ __forceinline__ void push(int sp, StackEntry v) { base[sp * stride] = v; }
  __device__ __forceinline__ StackEntry pop(int sp) const { return base[sp * stride]; }
};
__device__ __forceinline__ LdsStack lds_stack() { return LdsStack{g_lds_stack + threadIdx.x, (int)blockDim.x}; }

__device__ __forceinline__ PathState load_state(const RenderParams& p, uint32_t px) {
  const q4 a = p.st_o[px], b = p.st_d[px], c = p.st_s[px];
  const u4 m = p.st_m[px];
  PathState s;
  s.o = mk3(a.x, a.y, a.z);
  s.d = mk3(b.x, b.y, b.z);
  s.sum = mk3(c.x, c.y, c.z);
  s.T = mk3(a.w, b.w, c.w);
  s.seed = ((uint64_t)m.y << 32) | (uint64_t)m.x;
  s.prev_pdf = u2f(m.z);
  s.flags = m.w;
  s.origin = 0u;
  return s;
}
__device__ __forceinline__ void store_state(const RenderParams& p, uint32_t px, const PathState& s) {
  p.st_o[px] = q4{s.o.x, s.o.y, s.o.z, s.T.x};
  p.st_d[px] = q4{s.d.x, s.d.y, s.d.z, s.T.y};
  p.st_s[px] = q4{s.sum.x, s.sum.y, s.sum.z, s.T.z};
  p.st_m[px] = u4{(uint32_t)s.seed, (uint32_t)(s.seed >> 32), f2u(s.prev_pdf), s.flags};
}

// A slot id unpacked: which pixel, and which global sample range this slot renders.
struct Slot {
  uint32_t pixel;
  int x, y;
  int begin;  // global index of the slot's first sample
  int count;  // samples this slot takes
};
// a / b for a, b < 2^31 with inv = ceil(2^32 / b) made on the host (0 for b = 1, whose constant does not fit): the
// high half of a * inv is the quotient or one more
__device__ __forceinline__ uint32_t fast_div(uint32_t a, uint32_t b, uint32_t inv) {
  uint32_t q = __umulhi(a, inv);
  if (q * b > a) --q;
  return inv ? q : a;
}
__device__ __forceinline__ Slot decode_slot(const RenderParams& p, uint32_t slot) {
  Slot s;
  const uint32_t chunk = p.chunks > 1 ? fast_div(slot, p.npx, p.npx_inv) : 0u;
  s.pixel = slot - chunk * p.npx;
  s.y = (int)fast_div(s.pixel, (uint32_t)p.width, p.width_inv);
  s.x = (int)(s.pixel - (uint32_t)s.y * (uint32_t)p.width);
  const int c = (int)chunk, is_long = c < p.chunk_long;
  const int first = is_long ? c * p.chunk_spp : p.chunk_long * p.chunk_spp + (c - p.chunk_long) * p.chunk_short;
  const int len = is_long ? p.chunk_spp : p.chunk_short;
  s.begin = p.sample_begin + first;
  const int rem = p.sample_count - first;
  s.count = rem < len ? (rem > 0 ? rem : 0) : len;
  if (p.round_limit > 0) {
    const int limit = c < p.round_split ? p.round_limit : p.round_limit_b;
    if (s.count > limit) s.count = limit;
  }
  return s;
}
// Sample index fed to the jitter hash, and the PCG state the sample starts from.
// strict: optix_device.cu:945-948,953 (seed re-derived and s restarted every 4 samples);
// converge: global index, stream continues (re-seeded at a chunk start).
__device__ __forceinline__ int hash_sample_index(const RenderParams& p, const Slot& sl, int local) {
  return p.strict ? (local & 3) : sl.begin + local;
}
__device__ __forceinline__ uint64_t start_seed(const RenderParams& p, const Slot& sl, int local, uint64_t carried) {
  if (p.strict) return (local & 3) == 0 ? pixel_seed(p.frame, sl.pixel) : carried;
  return local == 0 ? chunk_seed(p.frame, sl.pixel, sl.begin) : carried;
}

// Append `keep` lanes' values to a queue: ballot, one atomic per wave, prefix popcount.
__device__ __forceinline__ uint32_t wave_append_slot(bool keep, uint32_t* counter) {
  const unsigned long long mask = __ballot(keep);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == 0 && mask) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, 0);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Adaptive sampling keeps the per-channel sum of squared sample values beside the sum.
__device__ __forceinline__ void add_squares(q4& a, f3 v) { a.x += v.x * v.x; a.y += v.y * v.y; a.z += v.z * v.z; }
__device__ __forceinline__ void fold_squares(q4* acc2, uint32_t slot, f3 v) {
  q4 a = acc2[slot];
  add_squares(a, v);
  acc2[slot] = a;
}

__device__ __forceinline__ ShadeParams shade_params(const RenderParams& p) {
  return ShadeParams{p.frame, p.max_depth, p.guiding};
}

__device__ __forceinline__ TrainRec train_rec(const TrainVertex& tv) {
  TrainRec r;
  r.position[0] = tv.P.x; r.position[1] = tv.P.y; r.position[2] = tv.P.z;
  r.dir_in[0] = tv.dir_in.x; r.dir_in[1] = tv.dir_in.y; r.dir_in[2] = tv.dir_in.z;
  r.contrib[0] = tv.T_in.x; r.contrib[1] = tv.T_in.y; r.contrib[2] = tv.T_in.z;
  r.is_delta = train_digest(r);  // flag bit 0 = 0: only non-delta vertices are captured
  return r;
}
// Training capture (optix_device.cu:431-441).  The reference appends every non-delta vertex to a ring
// through one atomic counter; a 1080p render makes thousands of times more vertices than the ring holds,
// and that counter then serialises the whole render (a returning same-address atomic per wave and bounce
// made the fused kernel 2.7x slower).  Here a vertex goes to the ring slot its own random state hashes
// to -- what survives is a uniform random subset of the render's vertices instead of whichever were
// appended last -- and the number of vertices is only counted (per lane, summed when the kernel ends).
// Two vertices that hash to one slot at the same moment may interleave their stores; the digest in
// is_delta lets the host tell.  Never-written slots keep the marker the ring is cleared to
// (is_delta == kTrainEmpty) and are dropped
// when the host maps the ring.
__device__ __forceinline__ void capture_train(const RenderParams& p, const TrainVertex& tv, uint64_t seed, uint32_t slot,
                                              uint32_t& captured) {
  if (!p.train || !tv.valid) return;
  captured++;
  if (!p.train_capacity) return;
  const uint32_t key = hash_u32((uint32_t)seed ^ (uint32_t)(seed >> 32) ^ (slot * 0x9E3779B1u));
  p.train[key % p.train_capacity] = train_rec(tv);
}
__device__ __forceinline__ void flush_train_count(const RenderParams& p, uint32_t captured) {
  for (int off = 32; off > 0; off >>= 1) captured += __shfl_down(captured, off);
  if ((threadIdx.x & 63u) == 0u && captured && p.train_write_idx) atomicAdd(p.train_write_idx, captured);
}

// Training statistics per (region, lobe) of the trainer's field, straight from the ring (pt_guiding.h).
__global__ void __launch_bounds__(256) k_bin_train(const TrainRec* ring, uint32_t capacity, BinField field, const float* axes,
                                                   unsigned long long* bins) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += gridDim.x * blockDim.x) {
    const TrainRec rec = ring[i];
    uint32_t bin;
    long long add[kBinWords];
    if (!bin_train_record(rec, field, axes, bin, add)) continue;
    // two's-complement addition: the unsigned atomic adds signed addends exactly
    for (int k = 0; k < kBinWords; ++k)
      if (add[k] != 0) atomicAdd(&bins[(size_t)bin * kBinWords + k], (unsigned long long)add[k]);
  }
}
void launch_bin_train(const TrainRec* ring, uint32_t capacity, const BinField& field, const float* d_axes, long long* d_bins,
                      hipStream_t stream) {
  if (capacity == 0) return;
  const int blocks = (int)std::min<uint32_t>((capacity + 255u) / 256u, 2048u);
  k_bin_train<<<blocks, 256, 0, stream>>>(ring, capacity, field, d_axes, reinterpret_cast<unsigned long long*>(d_bins));
  PT_KCHECK();
}

// ------------------------------------------------------------------------- generate
// The queue the staged kernels work through is kept in slot order: path state is found by slot number, and a
// wave whose 64 queue entries are 64 neighbouring slots reads and writes whole cache lines of it.  A compaction
// that appends the survivors of its blocks in the order the blocks happen to arrive keeps that only for a while
// (every pass cuts the runs of neighbours at its block borders: after a few hundred passes the queue is a
// shuffle, and k_shade needs nearly twice as long per path).  So the pass is stable: every tile of 1024 entries
// compacts its survivors in place into queue_tmp and records how many there are, a scan turns the counts into
// places, and a third kernel moves the tiles there.
constexpr uint32_t kTile = 1024;  // 4 waves x 4 rows of 64 consecutive entries

// Ranks of a block's kept items in (wave, item, lane) order, which is entry order when a wave holds ITEMS rows of
// 64 consecutive entries, lane by lane; returns how many are kept in all.
template <int ITEMS>
__device__ __forceinline__ uint32_t block_rank_slots(const bool (&keep)[ITEMS], uint32_t (&rank)[ITEMS]) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_total;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63u) >> 6;
  const unsigned long long lower = (1ull << lane) - 1ull;
  uint32_t in_wave = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const unsigned long long mask = __ballot(keep[k]);
    rank[k] = in_wave + (uint32_t)__popcll(mask & lower);
    in_wave += (uint32_t)__popcll(mask);
  }
  if (lane == 0) s_wave[wave] = in_wave;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < nwaves; ++w) { const uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
    s_total = total;
  }
  __syncthreads();
  const uint32_t base = s_wave[wave];
  const uint32_t total = s_total;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) rank[k] += base;
  __syncthreads();  // s_wave / s_total are reused by the next call
  return total;
}

__global__ void __launch_bounds__(256) k_generate(RenderParams p, int qin) {
  constexpr int ITEMS = 4;
  const uint32_t n = p.counters[qin];
  const uint32_t n_tiles = (n + kTile - 1u) / kTile;
  const uint32_t* q_in = p.queue[qin];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the traversal kernels that follow start at the front of their queues
    p.counters[CNT_EXTEND] = 0u;
    p.counters[CNT_OCCLUDE] = 0u;
  }
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t base = tile * kTile;
    bool keep[ITEMS];
    uint32_t pixel[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t i = base + (threadIdx.x >> 6) * (64u * ITEMS) + (uint32_t)k * 64u + (threadIdx.x & 63u);
      keep[k] = false;
      pixel[k] = 0;
      if (i >= n) continue;
      pixel[k] = q_in[i];
      const u4 m = p.st_m[pixel[k]];
      keep[k] = true;
      if (m.w & kAlive) continue;
      const int local = p.started[pixel[k]];
      if (m.w & kFoldPending) {  // a sample just ended: fold it into the slot's sum
        const q4 s = p.st_s[pixel[k]];
        q4 a = p.acc[pixel[k]];
        a.x += s.x; a.y += s.y; a.z += s.z;
        p.acc[pixel[k]] = a;
        if (p.acc2) fold_squares(p.acc2, pixel[k], mk3(s.x, s.y, s.z));
      }
      const Slot sl = decode_slot(p, pixel[k]);
      if (local >= sl.count) {
        keep[k] = false;
        // the slot may be queued again (next adaptive round): its last sample is folded only once
        if (m.w & kFoldPending) p.st_m[pixel[k]].w = m.w & ~kFoldPending;
      } else {
        const uint64_t carried = ((uint64_t)m.y << 32) | (uint64_t)m.x;
        PathState ps;
        generate_path(p.cam, make_rand_ctx(sl.x, sl.y, 0, p.frame), hash_sample_index(p, sl, local), p.width, p.height,
                      start_seed(p, sl, local, carried), ps);
        store_state(p, pixel[k], ps);
        p.started[pixel[k]] = local + 1;
      }
    }
    uint32_t rank[ITEMS];
    const uint32_t total = block_rank_slots<ITEMS>(keep, rank);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (keep[k]) p.queue_tmp[base + rank[k]] = pixel[k];
    if (threadIdx.x == 0) p.tiles[tile] = total;
  }
}
// counts per tile -> the tiles' places in the next queue (tiles[n_tiles] = its length, also put in its counter)
__global__ void __launch_bounds__(1024) k_tile_scan(RenderParams p, int qin) {
  constexpr uint32_t PER = 16;  // consecutive tiles per thread
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const uint32_t n_tiles = (p.counters[qin] + kTile - 1u) / kTile;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0u;
  __syncthreads();
  for (uint32_t start = 0; start < n_tiles; start += blockDim.x * PER) {
    const uint32_t first = start + threadIdx.x * PER;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      v[k] = first + k < n_tiles ? p.tiles[first + k] : 0u;
      sum += v[k];
    }
    uint32_t incl = sum;  // inclusive scan of the threads' sums over the wave ...
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      if ((int)lane >= off) incl += up;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = s_carry;  // ... and over the waves in front
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    uint32_t at = before + incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      if (first + k < n_tiles) p.tiles[first + k] = at;
      at += v[k];
    }
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1u) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    p.tiles[n_tiles] = s_carry;
    p.counters[qin ^ 1] = s_carry;
  }
}
__global__ void __launch_bounds__(256) k_tile_gather(RenderParams p, int qin) {
  const uint32_t n_tiles = (p.counters[qin] + kTile - 1u) / kTile;
  uint32_t* q_out = p.queue[qin ^ 1];
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t at = p.tiles[tile], count = p.tiles[tile + 1u] - at;
    for (uint32_t t = threadIdx.x; t < count; t += blockDim.x) q_out[at + t] = p.queue_tmp[tile * kTile + t];
  }
}

// --------------------------------------------------------------------------- extend
// Rays of very different length share a wave: with one ray per lane the wave runs as long as its
// longest ray while the other lanes idle (17 of 64 lanes busy on a 12.8 M-triangle scene).  Here
// every wave owns a contiguous range of the queue, and whenever refill_lanes lanes have finished
// their rays they take the next ones from the range while the rest keep traversing.  Each iteration
// advances every busy lane by one step: a child-node visit or a triangle test.

//
// Scenes whose rays are long (trees beyond the caches, RenderParams::shared_ranges) do not fix the ranges per
// wave: a wave that has used up what it held takes the next `grab` entries from a counter all waves share, so a wave
// whose rays were short goes on where the others have not been yet and the launch ends when the queue does, not
// when the wave with the slowest share does.  Where rays are cheap the same-address atomics would be the kernel
// (2.3 times slower on the built-in scene), and every wave gets one contiguous share beforehand.
struct QueueRange {
  uint32_t cursor, end;  // the same on every lane
  uint32_t grab;
  bool more;
};
__device__ __forceinline__ QueueRange first_range(uint32_t n, bool shared) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  QueueRange r;
  const uint32_t g = (n / (nwaves * 4u)) & ~63u;  // several ranges per wave, whole waves' worth of rays
  if (shared && g >= 512u) {  // shorter ranges cost more in atomics than the even finish gains
    r.grab = g > 2048u ? 2048u : g;
    r.cursor = r.end = 0u;
    r.more = true;
  } else {
    const uint32_t per = (n + nwaves - 1u) / nwaves;
    r.grab = 0u;
    r.cursor = wave * per;
    r.end = r.cursor + per < n ? r.cursor + per : n;
    r.more = false;
  }
  return r;
}
__device__ __forceinline__ void take_range(QueueRange& r, uint32_t* counter, uint32_t n) {
  uint32_t b = 0u;
  if ((threadIdx.x & 63u) == 0u) b = atomicAdd(counter, r.grab);
  b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
  const uint32_t grab = r.grab;
  if (b >= n) {
    r.more = false;
  } else {
    r.cursor = b;
    r.end = b + grab < n ? b + grab : n;
  }
}

// INST: the scene is made of instances (pt_bvh.h: TraceStateT<true>); the hit's instance goes to hit_inst.
template <bool INST>
__global__ void __launch_bounds__(256) k_extend(RenderParams p, int q) {
  const uint32_t n = p.counters[q];
  const uint32_t* queue = p.queue[q];
  LdsStack st = lds_stack();
  if (blockIdx.x == 0 && threadIdx.x == 0) p.ray_counters[0] += n;
  const uint32_t lane = threadIdx.x & 63u;
  QueueRange r = first_range(n, p.shared_ranges != 0);
  TraceStateT<INST> ts;
  ts.finish();
  f3 ro = mk3(0.f), rd = mk3(0.f);
  uint32_t idx = 0xffffffffu;  // queue position of the lane's ray, none yet
  prof_begin();
  for (;;) {
    const unsigned long long idle = __ballot(idx == 0xffffffffu);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    const bool refill = n_idle >= (uint32_t)p.refill_lanes || n_idle == 64u;
    if (refill && r.more && r.cursor >= r.end) take_range(r, &p.counters[CNT_EXTEND], n);
    if (r.cursor < r.end && refill) {
      if (idx == 0xffffffffu) {
        const uint32_t i = r.cursor + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        if (i < r.end) {
          idx = i;
          const uint32_t pixel = queue[i];
          const q4 a = p.st_o[pixel], b = p.st_d[pixel];
          ro = mk3(a.x, a.y, a.z);
          rd = mk3(b.x, b.y, b.z);
          ts.begin(p.scene, ro, rd, 1e16f, 0u);
        }
      }
      r.cursor += n_idle;
    } else if (n_idle == 64u) {
      break;
    }
    if (idx != 0xffffffffu) { PT_PROF(PF_BOUNCE); }
    if (idx != 0xffffffffu && ts.wants_node()) {
      PT_PROF(PF_NODE);
      ts.node_step(p.scene, ro, rd, 0.0f, st, p.stack_depth);
    }
    if (idx != 0xffffffffu && ts.wants_tri()) {
      PT_PROF(PF_TRI);
      ts.tri_step(p.scene, ro, rd, 0.0f, false, st, p.stack_depth);
    }
    if (idx != 0xffffffffu && ts.done()) {
      p.hits[idx] = q4{u2f((uint32_t)ts.hit.prim), ts.hit.t, ts.hit.b1, ts.hit.b2};
      if constexpr (INST) p.hit_inst[idx] = ts.hit_inst;
      idx = 0xffffffffu;
    }
  }
  prof_end();
}

// ---------------------------------------------------------------------------- shade
template <bool GUIDED, bool TRAIN, bool INST>
__global__ void __launch_bounds__(256) k_shade(RenderParams p, int q) {
  const uint32_t n = p.counters[q];
  const uint32_t* queue = p.queue[q];
  const ShadeParams sp = shade_params(p);
  uint32_t captured = 0;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < n;
    ShadowRay sr;
    sr.valid = false;
    TrainVertex tv;
    tv.valid = false;
    uint32_t pixel = 0;
    uint64_t seed_after = 0;
    if (valid) {
      pixel = queue[i];
      PathState ps = load_state(p, pixel);
      const q4 hq = p.hits[i];
      const Hit h{(int)f2u(hq.x), hq.y, hq.z, hq.w};
      const Slot sl = decode_slot(p, pixel);
      RandCtx rc = make_rand_ctx(sl.x, sl.y, 0, p.frame);
      rc.s = ps_dim(ps) < 2u ? hash_sample_index(p, sl, p.started[pixel] - 1) : 0;
      if constexpr (INST) shade_hit_instanced<GUIDED, TRAIN>(p.scene, p.guide, sp, rc, ps, h, p.hit_inst[i], sr, tv);
      else shade_hit<GUIDED, TRAIN>(p.scene, p.guide, sp, rc, ps, h, sr, tv);
      seed_after = ps.seed;
      store_state(p, pixel, ps);
    }
    if (TRAIN && valid) capture_train(p, tv, seed_after, pixel, captured);
    // About half of the paths make a shadow ray.  The records of a block's 256 paths go to the front of the block's
    // 256 places, in path order, and their number into sh_count (no atomic): the occlusion kernel then takes
    // shadow rays only, instead of finding out record by record that there is nothing to trace.
    {
      __shared__ uint32_t s_rays[4];
      const bool has = valid && sr.valid;
      const unsigned long long mask = __ballot(has);
      const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
      if (lane == 0u) s_rays[wave] = (uint32_t)__popcll(mask);
      __syncthreads();
      uint32_t before = 0u, total = 0u;
      for (uint32_t w = 0; w < 4u; ++w) {
        if (w < wave) before += s_rays[w];
        total += s_rays[w];
      }
      if (has) {
        const uint32_t j = base + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        p.sh_o[j] = q4{sr.o.x, sr.o.y, sr.o.z, sr.tmax};
        p.sh_d[j] = q4{sr.d.x, sr.d.y, sr.d.z, u2f(pixel)};
        p.sh_c[j] = q4{sr.contrib.x, sr.contrib.y, sr.contrib.z, u2f(sr.origin)};
      }
      if (threadIdx.x == 0u) p.sh_count[base >> 8] = total;
      __syncthreads();  // s_rays is reused by the next pass
    }
  }
  if (TRAIN) flush_train_count(p, captured);
}

// -------------------------------------------------------------------------- occlude
template <bool INST>
__global__ void __launch_bounds__(256) k_occlude(RenderParams p, int q) {
  const uint32_t n = p.counters[q];
  LdsStack st = lds_stack();
  if (blockIdx.x == 0 && threadIdx.x == 0) p.counters[q ^ 1] = 0u;  // the queue the next generate pass appends to
  const uint32_t lane = threadIdx.x & 63u;
  QueueRange r = first_range(n, p.shared_ranges != 0);
  uint32_t traced = 0;
  TraceStateT<INST> ts;
  ts.finish();
  f3 ro = mk3(0.f), rd = mk3(0.f);
  uint32_t idx = 0xffffffffu;
  prof_begin();
  for (;;) {  // same scheme as k_extend
    const unsigned long long idle = __ballot(idx == 0xffffffffu);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    const bool refill = n_idle >= (uint32_t)p.refill_lanes || n_idle == 64u;
    if (refill && r.more && r.cursor >= r.end) take_range(r, &p.counters[CNT_OCCLUDE], n);
    if (r.cursor < r.end && refill) {
      // shadow records sit at the front of every 256 places (k_shade): sh_count says how many
      const uint32_t have = p.sh_count[r.cursor >> 8], at = r.cursor & 255u;
      if (at >= have) {
        r.cursor = (r.cursor | 255u) + 1u;  // nothing left in these 256: on to the next
      } else {
        const uint32_t left = have - at < r.end - r.cursor ? have - at : r.end - r.cursor;
        const uint32_t take = n_idle < left ? n_idle : left;
        if (idx == 0xffffffffu) {
          const uint32_t k = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
          if (k < take) {
            const uint32_t i = r.cursor + k;
            idx = i;
            const q4 o = p.sh_o[i], d = p.sh_d[i];
            traced++;
            ro = mk3(o.x, o.y, o.z);
            rd = mk3(d.x, d.y, d.z);
            ts.begin(p.scene, ro, rd, o.w, f2u(p.sh_c[i].w));
          }
        }
        r.cursor += take;
      }
    } else if (n_idle == 64u) {
      break;
    }
    if (idx != 0xffffffffu) { PT_PROF(PF_SHADOW); }
    if (idx != 0xffffffffu && ts.wants_node()) {
      PT_PROF(PF_SH_NODE);
      ts.node_step(p.scene, ro, rd, 0.0f, st, p.stack_depth);
    }
    if (idx != 0xffffffffu && ts.wants_tri()) {
      PT_PROF(PF_SH_TRI);
      if (ts.tri_step(p.scene, ro, rd, 0.0f, true, st, p.stack_depth)) ts.finish();
    }
    if (idx != 0xffffffffu && ts.done()) {
      if (ts.hit.prim < 0) {
        const uint32_t pixel = f2u(p.sh_d[idx].w);
        const q4 c = p.sh_c[idx];
        q4 s = p.st_s[pixel];
        s.x += c.x; s.y += c.y; s.z += c.z;
        p.st_s[pixel] = s;
      }
      idx = 0xffffffffu;
    }
  }
  for (int off = 32; off > 0; off >>= 1) traced += __shfl_down(traced, off);
  if (lane == 0u && traced) atomicAdd(&p.ray_counters[1], (unsigned long long)traced);
  prof_end();
}

// ---------------------------------------------------------------------------- fused
// compact != 0: survivors are appended to the other queue (and the host swaps queues);
// compact == 0: the queue is processed in place -- finished slots stay in it, each idling a lane of its wave.
// Queue lengths live in three counters the launches rotate through (QueueCounters, pt_kernels.h).
#ifndef PT_BOUNCE_WAVES
#define PT_BOUNCE_WAVES 4
#endif
// Small scenes: wide nodes (pt_bvh.h), item records, shading records and the wide tree's home table are staged
// in LDS behind the traversal stacks, so the divergent 16-byte gathers of traversal and shading never leave
// the CU.  Returns the first free u4 after the staged scene.
__host__ __device__ inline uint32_t staged_scene_words(uint32_t nodes, uint32_t items, uint32_t tris) {
  return nodes * kWideNodeWords + items * (uint32_t)kTriWords + tris * 6u + (2u * tris + 3u) / 4u;
}
__device__ __forceinline__ u4* stage_scene(const RenderParams& p, u4* dst, SceneView& sc) {
  const uint32_t n_nodes16 = p.scene_nodes * kWideNodeWords, n_tris16 = p.scene.num_items * (uint32_t)kTriWords,
                 n_shade16 = p.scene.num_tris * 6u, n_home = 2u * p.scene.num_tris;
  const u4* src_t = reinterpret_cast<const u4*>(p.scene.tris);
  const u4* src_s = reinterpret_cast<const u4*>(p.scene.shade);
  for (uint32_t k = threadIdx.x; k < n_nodes16; k += blockDim.x) dst[k] = p.scene.wide_src[k];
  for (uint32_t k = threadIdx.x; k < n_tris16; k += blockDim.x) dst[n_nodes16 + k] = src_t[k];
  for (uint32_t k = threadIdx.x; k < n_shade16; k += blockDim.x) dst[n_nodes16 + n_tris16 + k] = src_s[k];
  uint32_t* home = reinterpret_cast<uint32_t*>(dst + n_nodes16 + n_tris16 + n_shade16);
  for (uint32_t k = threadIdx.x; k < n_home; k += blockDim.x) home[k] = p.scene.home_wide[k];
  __syncthreads();
  sc.wide = dst;
  sc.tris = reinterpret_cast<const q4*>(dst + n_nodes16);
  sc.shade = reinterpret_cast<const ShadeRec*>(dst + n_nodes16 + n_tris16);
  sc.home_wide = home;
  return dst + staged_scene_words(p.scene_nodes, p.scene.num_items, p.scene.num_tris);
}

// The camera -- 12 numbers a lane needs once per sample -- waits in LDS, at the end of the block's allocation, rather
// than in scalar registers, of which the bounce loop has too few: the compiler parks what does not fit in a vector
// register and fetches it back lane by lane (v_readlane) inside the loop.  With the camera out of the way the
// one-launch kernel spills 10 scalar registers instead of 33 and runs 1.4 % faster.
constexpr uint32_t kCamLdsBytes = 64;
constexpr uint32_t kTotalsWords = 8;  // k_bounce: a slot's sum and sum of squares (two q4) per thread, in LDS
__device__ __forceinline__ const float* stage_camera(const RenderParams& p, void* at) {  // whole block; ends with a barrier
  float* c = reinterpret_cast<float*>(at);
  if (threadIdx.x < 12u) c[threadIdx.x] = reinterpret_cast<const float*>(&p.cam)[threadIdx.x];
  __syncthreads();
  return c;
}
__device__ __forceinline__ CameraView lds_camera(const float* c) {
  CameraView cam;
  for (int k = 0; k < 3; ++k) {
    cam.eye[k] = c[k]; cam.u[k] = c[3 + k]; cam.v[k] = c[6 + k]; cam.w[k] = c[9 + k];
  }
  return cam;
}

template <bool GUIDED, bool TRAIN, bool LDS_SCENE>
__global__ void __launch_bounds__(256, PT_BOUNCE_WAVES) k_bounce(RenderParams p, int qin, int bounces, int compact, QueueCounters cnt) {
  const uint32_t n = p.counters[cnt.in];
  const uint32_t* q_in = p.queue[qin];
  uint32_t* q_out = p.queue[qin ^ 1];
  LdsStack st = lds_stack();
  SceneView sc = p.scene;
  u4* lds_free = reinterpret_cast<u4*>(g_lds_stack + (size_t)blockDim.x * (size_t)p.stack_depth);
  if (LDS_SCENE) {
    lds_free = stage_scene(p, lds_free, sc);
    __builtin_assume(sc.wide != nullptr);
  } else {
    sc.wide = nullptr;
  }
  const float* cam_lds = stage_camera(p, lds_free);
  // The slot's totals (sum, and sum of squares in adaptive renders) wait in LDS while the launch runs: a finished
  // sample is added there, in the order and with the rounding it would be added to the arrays in memory, and the
  // totals go back once, with the path state -- not a load, a wait and a store per sample inside the bounce loop.
  float* tot = reinterpret_cast<float*>(lds_free) + kCamLdsBytes / sizeof(float) + threadIdx.x;  // [8][blockDim]
  const uint32_t ts = blockDim.x;
#ifdef PT_PROFILE
  if (threadIdx.x < 64) g_prof_lds[threadIdx.x] = 0u;
  __syncthreads();
#endif
  const ShadeParams sp = shade_params(p);
  uint32_t rays_rad = 0, rays_sh = 0, captured = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) p.counters[cnt.clear] = 0u;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < n;
    bool keep = false;
    uint32_t pixel = 0;
    if (valid) {
      pixel = q_in[i];
      PathState ps = load_state(p, pixel);
      int local = p.started[pixel];
      const Slot sl = decode_slot(p, pixel);
      keep = (ps.flags & (kAlive | kFoldPending)) || local < sl.count;
      if (keep) {
        {
          const q4 a = p.acc[pixel];
          tot[0] = a.x; tot[ts] = a.y; tot[2u * ts] = a.z; tot[3u * ts] = a.w;
          if (p.acc2) {
            const q4 b = p.acc2[pixel];
            tot[4u * ts] = b.x; tot[5u * ts] = b.y; tot[6u * ts] = b.z; tot[7u * ts] = b.w;
          }
        }
        RandCtx rc = make_rand_ctx(sl.x, sl.y, 0, p.frame);  // the pixel's noise shift: once per launch
        for (int b = 0; b < bounces; ++b) {
          PT_PROF(PF_BOUNCE);
          if (!(ps.flags & kAlive)) {
            PT_PROF(PF_REGEN);
            if (ps.flags & kFoldPending) {  // a sample just ended: fold it into the slot's totals
              tot[0] += ps.sum.x; tot[ts] += ps.sum.y; tot[2u * ts] += ps.sum.z;
              if (p.acc2) {
                q4 b = q4{tot[4u * ts], tot[5u * ts], tot[6u * ts], 0.0f};
                add_squares(b, ps.sum);
                tot[4u * ts] = b.x; tot[5u * ts] = b.y; tot[6u * ts] = b.z;
              }
              ps.flags &= ~kFoldPending;
            }
            if (local >= sl.count) { keep = false; break; }
            generate_path(lds_camera(cam_lds), rc, hash_sample_index(p, sl, local), p.width, p.height,
                          start_seed(p, sl, local, ps.seed), ps);
            local++;
          }
          rc.s = hash_sample_index(p, sl, local - 1);
          const Hit h = trace_ray(sc, ps.o, ps.d, 0.0f, 1e16f, false, st, p.stack_depth, ps.origin);
          rays_rad++;
          ShadowRay sr;
          TrainVertex tv;
          shade_hit<GUIDED, TRAIN>(sc, p.guide, sp, rc, ps, h, sr, tv);
          if (TRAIN) capture_train(p, tv, ps.seed, pixel, captured);
          if (sr.valid) {
            PT_PROF(PF_SHADOW);
            rays_sh++;
            const Hit o = trace_ray(sc, sr.o, sr.d, 0.0f, sr.tmax, true, st, p.stack_depth, sr.origin);
            if (o.prim < 0) ps.sum += sr.contrib;
          }
        }
        // a path that ended on the last bounce of this launch is folded by the next launch
        store_state(p, pixel, ps);
        p.started[pixel] = local;
        p.acc[pixel] = q4{tot[0], tot[ts], tot[2u * ts], tot[3u * ts]};
        if (p.acc2) p.acc2[pixel] = q4{tot[4u * ts], tot[5u * ts], tot[6u * ts], tot[7u * ts]};
      }
    }
    if (compact) {
      const uint32_t slot = wave_append_slot(keep, &p.counters[cnt.out]);
      if (keep) q_out[slot] = pixel;
    }
  }
#ifdef PT_PROFILE
  __syncthreads();
  if (threadIdx.x < 64 && g_prof_lds[threadIdx.x]) atomicAdd(&g_prof_global[threadIdx.x], (unsigned long long)g_prof_lds[threadIdx.x]);
#endif
  if (TRAIN) flush_train_count(p, captured);
  // ray counters: wave reduction, one 64-bit atomic pair per wave
  for (int off = 32; off > 0; off >>= 1) {
    rays_rad += __shfl_down(rays_rad, off);
    rays_sh += __shfl_down(rays_sh, off);
  }
  if ((threadIdx.x & 63u) == 0u) {
    if (rays_rad) atomicAdd(&p.ray_counters[0], (unsigned long long)rays_rad);
    if (rays_sh) atomicAdd(&p.ray_counters[1], (unsigned long long)rays_sh);
  }
}

// ------------------------------------------------------------- fused, shadow rays deferred
// Same path loop as k_bounce, but a bounce's shadow ray is not traced where it is made.  About 45 % of
// the lanes of a wave have one per bounce, so tracing them in place runs the any-hit traversal at
// under half occupancy of the wave.  Instead the lanes append their shadow rays to a queue the wave
// keeps in LDS, and whenever 64 are waiting the whole wave traces them, one per lane.  An unoccluded
// ray's contribution goes back to the lane that owns the path through an LDS accumulator and is added
// to that slot's running sum at the end of the launch (or to the slot's total if its last sample has
// already been folded).  A slot's total is therefore the same set of terms as in k_bounce, added in
// a different order; per-sample sums are not preserved, so adaptive renders use k_bounce.  (A variant that kept them
// -- two accumulator sets per lane, a sample parked until its last queue entry had been traced -- was built and
// measured: bit-reproducible, 8 % fewer vector instructions than k_bounce, and 10 % slower: 48 instead of 19 spilled
// scalar registers, read back lane by lane inside the bounce loop.)
constexpr uint32_t kDeferCap = 128;                       // queue entries per wave (fewer than 64 wait, up to 64 arrive)
constexpr uint32_t kDeferFields = 12;                     // o.xyz d.xyz tmax origin contrib.xyz owner
constexpr uint32_t kDeferSet = 3u * 64u;                  // per-lane accumulators (x, y, z)
constexpr uint32_t kDeferWords = kDeferFields * kDeferCap + kDeferSet;

template <bool GUIDED, bool TRAIN, bool LDS_SCENE>
__global__ void __launch_bounds__(256, PT_BOUNCE_WAVES) k_bounce_defer(RenderParams p, int qin, int bounces, int compact,
                                                                       QueueCounters cnt) {
  const uint32_t n = p.counters[cnt.in];
  const uint32_t* q_in = p.queue[qin];
  uint32_t* q_out = p.queue[qin ^ 1];
  LdsStack st = lds_stack();
  SceneView sc = p.scene;
  u4* lds_free = reinterpret_cast<u4*>(g_lds_stack + (size_t)blockDim.x * (size_t)p.stack_depth);
  if (LDS_SCENE) {
    lds_free = stage_scene(p, lds_free, sc);
    __builtin_assume(sc.wide != nullptr);
  } else {
    sc.wide = nullptr;
  }
#ifdef PT_PROFILE
  if (threadIdx.x < 64) g_prof_lds[threadIdx.x] = 0u;
  __syncthreads();
#endif
  const uint32_t lane = threadIdx.x & 63u;
  float* dq = reinterpret_cast<float*>(lds_free) + (size_t)(threadIdx.x >> 6) * kDeferWords;  // this wave's queue
  float* late = dq + kDeferFields * kDeferCap;                                                // [3][64]
  const float* cam_lds = stage_camera(p, reinterpret_cast<float*>(lds_free) + (size_t)(blockDim.x >> 6) * kDeferWords);
  const ShadeParams sp = shade_params(p);
  uint32_t rays_rad = 0, rays_sh = 0, captured = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) p.counters[cnt.clear] = 0u;

  // trace queue entry e on this lane; credit its owner if the light is visible
  auto trace_deferred = [&](uint32_t e) {
    const f3 o = mk3(dq[0 * kDeferCap + e], dq[1 * kDeferCap + e], dq[2 * kDeferCap + e]);
    const f3 d = mk3(dq[3 * kDeferCap + e], dq[4 * kDeferCap + e], dq[5 * kDeferCap + e]);
    const float tmax = dq[6 * kDeferCap + e];
    const uint32_t origin = f2u(dq[7 * kDeferCap + e]);
    rays_sh++;
    PT_PROF(PF_SHADOW);
    PT_TIC(t_sh);
    const Hit h = trace_ray(sc, o, d, 0.0f, tmax, true, st, p.stack_depth, origin);
    if (h.prim < 0) {
      const uint32_t owner = f2u(dq[11 * kDeferCap + e]);
      atomicAdd(&late[owner], dq[8 * kDeferCap + e]);
      atomicAdd(&late[64u + owner], dq[9 * kDeferCap + e]);
      atomicAdd(&late[128u + owner], dq[10 * kDeferCap + e]);
    }
    PT_TOC(t_sh, PT_T_SHADOW);
  };

  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < n;
    bool keep = false, entered = false;
    uint32_t pixel = 0;
    int local = 0;
    PathState ps;
    ps.flags = 0u;
    Slot sl;
    RandCtx rc;
    if (valid) {
      pixel = q_in[i];
      ps = load_state(p, pixel);
      local = p.started[pixel];
      sl = decode_slot(p, pixel);
      keep = (ps.flags & (kAlive | kFoldPending)) || local < sl.count;
      entered = keep;
      rc = make_rand_ctx(sl.x, sl.y, 0, p.frame);  // the pixel's noise shift: once per launch, not per draw
    }
    late[lane] = 0.0f; late[64u + lane] = 0.0f; late[128u + lane] = 0.0f;
    bool active = keep;
    // ring buffer, first in first out: a path's contributions are then added to `late` in the order
    // its bounces made them, whatever else shares the wave, so a slot's result does not depend on
    // which slots it is grouped with (queue order after a compaction is not reproducible)
    uint32_t head = 0, waiting = 0;  // the same on every lane
    PT_TIC(t_loop);
    for (int b = 0; b < bounces; ++b) {
      ShadowRay sr;
      sr.valid = false;
      if (active) { PT_PROF(PF_BOUNCE); }
      PT_TIC(t_regen);
      if (active && !(ps.flags & kAlive)) {
        PT_PROF(PF_REGEN);
        if (ps.flags & kFoldPending) {  // a sample just ended: fold it into the slot's sum
          q4 acc = p.acc[pixel];
          acc.x += ps.sum.x; acc.y += ps.sum.y; acc.z += ps.sum.z;
          p.acc[pixel] = acc;
          ps.flags &= ~kFoldPending;
        }
        if (local >= sl.count) {
          keep = false;
          active = false;
        } else {
          generate_path(lds_camera(cam_lds), rc, hash_sample_index(p, sl, local), p.width, p.height,
                        start_seed(p, sl, local, ps.seed), ps);
          local++;
        }
      }
      PT_TOC(t_regen, PT_T_REGEN);
      if (active) {
        rc.s = hash_sample_index(p, sl, local - 1);
        PT_TIC(t_trace);
        const Hit h = trace_ray(sc, ps.o, ps.d, 0.0f, 1e16f, false, st, p.stack_depth, ps.origin);
        PT_TOC(t_trace, PT_T_TRACE);
        rays_rad++;
        TrainVertex tv;
        PT_TIC(t_shade);
        shade_hit<GUIDED, TRAIN>(sc, p.guide, sp, rc, ps, h, sr, tv);
        if (TRAIN) capture_train(p, tv, ps.seed, pixel, captured);
        PT_TOC(t_shade, PT_T_SHADE);
      }
      PT_TIC(t_enq);
      const unsigned long long made = __ballot(sr.valid);
      if (sr.valid) {
        const uint32_t e = (head + waiting + (uint32_t)__popcll(made & ((1ull << lane) - 1ull))) & (kDeferCap - 1u);
        dq[0 * kDeferCap + e] = sr.o.x; dq[1 * kDeferCap + e] = sr.o.y; dq[2 * kDeferCap + e] = sr.o.z;
        dq[3 * kDeferCap + e] = sr.d.x; dq[4 * kDeferCap + e] = sr.d.y; dq[5 * kDeferCap + e] = sr.d.z;
        dq[6 * kDeferCap + e] = sr.tmax;
        dq[7 * kDeferCap + e] = u2f(sr.origin);
        dq[8 * kDeferCap + e] = sr.contrib.x; dq[9 * kDeferCap + e] = sr.contrib.y; dq[10 * kDeferCap + e] = sr.contrib.z;
        dq[11 * kDeferCap + e] = u2f(lane);
      }
      waiting += (uint32_t)__popcll(made);
      PT_TOC(t_enq, PT_T_ENQUEUE);
      if (waiting >= 64u) {
        trace_deferred((head + lane) & (kDeferCap - 1u));
        head = (head + 64u) & (kDeferCap - 1u);
        waiting -= 64u;
      }
      if (__ballot(active) == 0ull) break;
    }
    PT_TOC(t_loop, PT_T_LOOP);
    PT_TIC(t_epi);
    if (lane < waiting) trace_deferred((head + lane) & (kDeferCap - 1u));  // flush what is left, at whatever occupancy
    if (entered) {
      const f3 add = mk3(late[lane], late[64u + lane], late[128u + lane]);
      if (keep) {
        ps.sum += add;  // lands in the running sample, or in the one waiting to be folded
      } else if (add.x != 0.0f || add.y != 0.0f || add.z != 0.0f) {
        q4 acc = p.acc[pixel];  // the slot's last sample is already folded
        acc.x += add.x; acc.y += add.y; acc.z += add.z;
        p.acc[pixel] = acc;
      }
      store_state(p, pixel, ps);
      p.started[pixel] = local;
    }
    if (compact) {
      const uint32_t slot = wave_append_slot(keep, &p.counters[cnt.out]);
      if (keep) q_out[slot] = pixel;
    }
    PT_TOC(t_epi, PT_T_EPILOGUE);
  }
#ifdef PT_PROFILE
  __syncthreads();
  if (threadIdx.x < 64 && g_prof_lds[threadIdx.x]) atomicAdd(&g_prof_global[threadIdx.x], (unsigned long long)g_prof_lds[threadIdx.x]);
#endif
  if (TRAIN) flush_train_count(p, captured);
  for (int off = 32; off > 0; off >>= 1) {
    rays_rad += __shfl_down(rays_rad, off);
    rays_sh += __shfl_down(rays_sh, off);
  }
  if (lane == 0u) {
    if (rays_rad) atomicAdd(&p.ray_counters[0], (unsigned long long)rays_rad);
    if (rays_sh) atomicAdd(&p.ray_counters[1], (unsigned long long)rays_sh);
  }
}

// ------------------------------------------------------------- fused, one launch per queue
// The kernels above work through the slot queue launch after launch: a launch gives every slot a lane for a fixed
// number of bounces, a slot that runs out of samples idles its lane until the launch ends (about 7 % of all lane
// time: a slot lives for a dozen launches and ends, on average, half-way through one), and every launch ends with
// its slowest wave.  Here the whole queue is one launch.  The grid is what the GPU keeps resident; every wave takes
// queue entries in stretches of 64 from a counter all waves share and hands them to its lanes one by one: a lane
// whose slot has taken its last sample writes the slot back and takes the next entry in the same iteration, so the
// lanes stay busy until the queue is empty.  Nothing of a path touches memory between its slot's first and last
// bounce, and there is no compaction: when the launch ends every slot of the queue has finished.
//
// !DEFER (the default): shadow rays are traced where they are made and a finished sample is added to the slot's
// sum, which waits in LDS (three words per lane) and starts from what the slot held before: the additions of
// k_bounce and of the staged kernels in their order, so all three render the same bits.
// DEFER: shadow rays go through the wave's queue in LDS as in k_bounce_defer.  The lane keeps two accumulators in
// LDS for as long as its slot lives: the sum of the slot's finished samples, in sample order, and the credits of its
// unoccluded shadow rays, in queue order; their sum goes to the slot's total when it retires, which it does only
// after its last queued ray has been traced (two counters per wave -- entries queued, entries traced -- and the
// position of the lane's last entry tell when; the wait is two iterations out of hundreds).
// Either way a slot's result does not depend on the lane or the wave that carried it, or on what shared the wave:
// renders stay reproducible bit for bit although slots reach lanes in a different order every time.
//
// Item tests shared across the wave were built on this kernel and taken out again.  A lane tested its ray's first
// candidate itself; the candidates left over in the wave went to a list in LDS and were tested 64 pairs at a time by
// whichever lane, the ray read from its owner's registers (ds_bpermute), the closest hit handed back through a
// 64-bit LDS minimum of (distance, testing lane).  It did what it was meant to -- 2.0 rounds of item tests per
// closest-hit query instead of 4.2, 58 % of lanes active instead of 54 % -- and was 13 % slower (412.8 against
// 366.0 ms): building the list, eleven lane reads and the key traffic cost more instructions than the two rounds of
// a 55-instruction test they replaced (+3.7 % vector, +9 % scalar, +10 % LDS instructions, 23 spilled registers
// against 7), and 27 of 10.8 G rays found another hit (a candidate is rejected against the closest hit before the
// node's items, not the closest so far, which matters where two surfaces meet within the rounding of one test).
//
// What it costs: a lane refills alone, so the hundred-odd instructions that write one slot back and set up the next
// run at one lane in 64 where a launch of k_bounce_defer does them for 64 slots at once (+3.3 % vector instructions
// measured, against 6.3 % fewer loop iterations).  With shadow rays in place the one launch is 4 % faster than launch
// after launch at 1024 samples a pixel (324.7 against 338.1 ms), with the queue 1-3 % (more on short renders, whose
// launches are all head and tail).  The rounds of an adaptive render pick slots up again: their stored state is
// fetched 32 slots at a time into LDS (the stash, below) -- fetched by the lane that takes the slot, one lane's trip
// to memory that the whole wave waits for, the one launch gained nothing there (345.6 against 345.8 ms); with the
// stash it is 3.6 % faster than launch after launch (335.0 against 347.1 ms at threshold 0.01), short rounds of 8
// samples a slot included.  (Sized for 64 slots at a time the stash took the fourth block off the CU: 387 ms.)
//
// RenderParams::fresh_queue: every slot of the queue is untouched (a fixed-count render's only launch):
// nothing is loaded for it.  RenderParams::last_pass: no slot will run again after this launch: only its totals are
// written.  Renders with a noise test keep the sum of squared samples beside the sum, both in LDS while the slot
// has a lane.

constexpr uint32_t kPersistGrab = 64;  // queue entries a wave takes from the shared counter at a time
// per wave: finished-sample sums; DEFER: the shadow queue (owner and origin share a word: 11 fields), credits,
// finished-sample sums, last-entry marks
__host__ __device__ constexpr uint32_t persist_words(bool defer) {
  return defer ? (kDeferFields - 1u) * kDeferCap + 2u * kDeferSet + 64u : 2u * kDeferSet;
}
// Queues whose slots have been worked on before (the rounds of an adaptive render, a staged render's tail): a wave
// fetches the stored state of the 64 entries it takes from the queue at once, every lane one entry, into LDS --
// six 16-byte words and the slot id per entry -- and a lane that takes a slot later reads it from there: one trip
// to memory per 64 slots for the wave to wait for, not one per slot.
constexpr uint32_t kStashEntries = 32;  // (such queues are taken 32 entries at a time: 64 would cost the fourth block per CU)
constexpr uint32_t kStashWords = kStashEntries * (6u * 4u + 1u);
constexpr uint32_t kPersistOriginBits = 26;  // a shadow ray's origin primitive (+1) beside its owner lane in one word

// SCENE: 0 the packed tree in device memory, 1 the wide tree in LDS, 2 the same when it is one node (trace_ray_one_node)
template <bool GUIDED, bool TRAIN, int SCENE, bool DEFER>
__global__ void __launch_bounds__(256, PT_BOUNCE_WAVES) k_bounce_persist(RenderParams p, int qin, int cnt_in) {
  constexpr bool LDS_SCENE = SCENE != 0;
  const uint32_t n = p.counters[cnt_in];
  const uint32_t* q_in = p.queue[qin];
  LdsStack st = lds_stack();
  SceneView sc = p.scene;
  u4* lds_free = reinterpret_cast<u4*>(g_lds_stack + (size_t)blockDim.x * (size_t)p.stack_depth);
  if (LDS_SCENE) {
    lds_free = stage_scene(p, lds_free, sc);
    __builtin_assume(sc.wide != nullptr);
  } else {
    sc.wide = nullptr;
  }
#ifdef PT_PROFILE
  if (threadIdx.x < 64) g_prof_lds[threadIdx.x] = 0u;
  __syncthreads();
#endif
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  float* dq = reinterpret_cast<float*>(lds_free) + (size_t)(threadIdx.x >> 6) * persist_words(DEFER);  // this wave's queue
  float* late = dq + (kDeferFields - 1u) * kDeferCap;                       // [3][64]: credits of the lane's slot
  float* done_sum = DEFER ? late + kDeferSet : dq;                          // [3][64]: its finished samples
  uint32_t* last_entry = reinterpret_cast<uint32_t*>(done_sum + kDeferSet); // [64]: 1 + position of its last queued ray
  float* sq_sum = done_sum + kDeferSet;  // !DEFER, renders with a noise test: [3][64] the slot's sum of squared samples
  const ShadeParams sp = shade_params(p);
  uint32_t rays_rad = 0, rays_sh = 0, captured = 0;
  const bool fresh = p.fresh_queue != 0, last_pass = p.last_pass != 0;
  const float* cam_lds = stage_camera(p, reinterpret_cast<float*>(lds_free) + (size_t)(blockDim.x >> 6) * persist_words(DEFER));
  // the stash of fetched slot state (queues that are not fresh), behind the camera: [6][64] q4 and [64] slot ids per wave
  q4* stash = reinterpret_cast<q4*>(const_cast<float*>(cam_lds) + kCamLdsBytes / sizeof(float)) + (size_t)(threadIdx.x >> 6) * (kStashWords / 4u);
  uint32_t* stash_px = reinterpret_cast<uint32_t*>(stash + 6u * kStashEntries);
  const uint32_t grab = fresh ? kPersistGrab : kStashEntries;

  auto trace = [&](f3 o, f3 d, float tmax, bool any_hit, uint32_t origin) -> Hit {
    if (SCENE == 2) {
      if (!any_hit) { PT_PROF(PF_NODE); } else { PT_PROF(PF_SH_NODE); }
      return trace_ray_one_node(sc, o, d, tmax, any_hit, origin, !any_hit);  // radiance rays start with tmax 1e16
    }
    return trace_ray(sc, o, d, 0.0f, tmax, any_hit, st, p.stack_depth, origin);
  };
  auto trace_deferred = [&](uint32_t e) {
    const f3 o = mk3(dq[0 * kDeferCap + e], dq[1 * kDeferCap + e], dq[2 * kDeferCap + e]);
    const f3 d = mk3(dq[3 * kDeferCap + e], dq[4 * kDeferCap + e], dq[5 * kDeferCap + e]);
    const float tmax = dq[6 * kDeferCap + e];
    const uint32_t word = f2u(dq[7 * kDeferCap + e]);
    rays_sh++;
    PT_PROF(PF_SHADOW);
    PT_TIC(t_sh);
    const Hit h = trace(o, d, tmax, true, word & ((1u << kPersistOriginBits) - 1u));
    if (h.prim < 0) {
      const uint32_t owner = word >> kPersistOriginBits;
      atomicAdd(&late[owner], dq[8 * kDeferCap + e]);
      atomicAdd(&late[64u + owner], dq[9 * kDeferCap + e]);
      atomicAdd(&late[128u + owner], dq[10 * kDeferCap + e]);
    }
    PT_TOC(t_sh, PT_T_SHADOW);
  };

  // the lane's slot
  bool has_slot = false, retiring = false;
  uint32_t pixel = 0;
  int local = 0;
  PathState ps;
  ps.o = ps.d = ps.T = ps.sum = mk3(0.0f);
  ps.seed = 0;
  ps.prev_pdf = 0.0f;
  ps.flags = 0u;
  ps.origin = 0u;
  Slot sl = decode_slot(p, 0u);
  RandCtx rc = make_rand_ctx(0, 0, 0, p.frame);
  if (DEFER) {
    late[lane] = 0.0f; late[64u + lane] = 0.0f; late[128u + lane] = 0.0f;
    last_entry[lane] = 0u;
  }
  done_sum[lane] = 0.0f; done_sum[64u + lane] = 0.0f; done_sum[128u + lane] = 0.0f;
  const bool squares = !DEFER && p.acc2 != nullptr;
  if (squares) { sq_sum[lane] = 0.0f; sq_sum[64u + lane] = 0.0f; sq_sum[128u + lane] = 0.0f; }
  // the same on every lane of the wave
  uint32_t head = 0, waiting = 0, queued_total = 0, traced_total = 0;  // the LDS queue
  uint32_t grab_pos = 0, grab_end = 0, grab_base = 0;                   // the stretch of the slot queue the wave holds
  bool more = n > 0u;                                                   // the shared counter has not run past the queue yet
#ifdef PT_PROFILE
  // when the first wave starts, when the first and the last wave find the queue empty, when the last wave ends
  // (100 MHz clock; minima kept as maxima of 2^62 - t): words 56..59 of the counters
  if (lane == 0u) atomicMax(&g_prof_global[56], (1ull << 62) - (unsigned long long)wall_clock64());
#endif
  PT_TIC(t_loop);
  for (;;) {
    // a slot whose last sample is folded and whose queued rays are all traced goes back to memory
    if (has_slot && retiring && (!DEFER || traced_total >= last_entry[lane])) {
      // DEFER: total = what the slot held before + its samples of this launch in order + their credits in queue order;
      // else the samples were added, in order, to what the slot held before
      f3 v = mk3(done_sum[lane], done_sum[64u + lane], done_sum[128u + lane]);
      q4 acc = q4{0.0f, 0.0f, 0.0f, 0.0f};
      if (DEFER) {
        v += mk3(late[lane], late[64u + lane], late[128u + lane]);
        if (!fresh) acc = p.acc[pixel];
        acc.x += v.x; acc.y += v.y; acc.z += v.z;
        late[lane] = 0.0f; late[64u + lane] = 0.0f; late[128u + lane] = 0.0f;
        last_entry[lane] = 0u;
      } else {
        acc.x = v.x; acc.y = v.y; acc.z = v.z;
      }
      p.acc[pixel] = acc;
      done_sum[lane] = 0.0f; done_sum[64u + lane] = 0.0f; done_sum[128u + lane] = 0.0f;
      if (squares) {
        p.acc2[pixel] = q4{sq_sum[lane], sq_sum[64u + lane], sq_sum[128u + lane], 0.0f};
        sq_sum[lane] = 0.0f; sq_sum[64u + lane] = 0.0f; sq_sum[128u + lane] = 0.0f;
      }
      if (!last_pass) store_state(p, pixel, ps);
      p.started[pixel] = local;
      has_slot = false;
      retiring = false;
    }
    // lanes without a slot take the next queue entries
    {
      unsigned long long need = __ballot(!has_slot);
      while (need != 0ull && (grab_pos < grab_end || more)) {
        if (grab_pos == grab_end) {  // a new stretch from the counter all waves share
          uint32_t b = 0u;
          if (lane == 0u) b = atomicAdd(&p.counters[CNT_GRAB], grab);
          b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
          if (b >= n) {
#ifdef PT_PROFILE
            if (lane == 0u) {
              const unsigned long long t = (unsigned long long)wall_clock64();
              atomicMax(&g_prof_global[57], (1ull << 62) - t);
              atomicMax(&g_prof_global[58], t);
            }
#endif
            more = false;
            break;
          }
          grab_pos = grab_base = b;
          grab_end = b + grab < n ? b + grab : n;
          if (!fresh) {
            if (b + lane < grab_end) {
              const uint32_t px = q_in[b + lane];
              stash_px[lane] = px;
              stash[lane] = p.st_o[px];
              stash[kStashEntries + lane] = p.st_d[px];
              stash[2u * kStashEntries + lane] = p.st_s[px];
              const u4 m = p.st_m[px];
              stash[3u * kStashEntries + lane] = q4{u2f(m.x), u2f(m.y), u2f(m.z), u2f(m.w)};
              q4 a = p.acc[px];
              a.w = u2f((uint32_t)p.started[px]);
              stash[4u * kStashEntries + lane] = a;
              if (squares) stash[5u * kStashEntries + lane] = p.acc2[px];
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
        const uint32_t n_need = (uint32_t)__popcll(need), room = grab_end - grab_pos;
        const uint32_t take = n_need < room ? n_need : room;
        const uint32_t rank = (uint32_t)__popcll(need & lanes_below);
        if (!has_slot && rank < take) {
          if (fresh) {  // an untouched slot: no samples started, no path under way
            pixel = q_in[grab_pos + rank];
            ps.flags = 0u;
            ps.seed = 0;
            local = 0;
          } else {  // what the wave fetched when it took this stretch of the queue
            const uint32_t e = grab_pos + rank - grab_base;
            pixel = stash_px[e];
            const q4 so = stash[e], sd = stash[kStashEntries + e], ss = stash[2u * kStashEntries + e], sm = stash[3u * kStashEntries + e], sa = stash[4u * kStashEntries + e];
            ps.o = mk3(so.x, so.y, so.z);
            ps.d = mk3(sd.x, sd.y, sd.z);
            ps.sum = mk3(ss.x, ss.y, ss.z);
            ps.T = mk3(so.w, sd.w, ss.w);
            ps.seed = ((uint64_t)f2u(sm.y) << 32) | (uint64_t)f2u(sm.x);
            ps.prev_pdf = sm.z;
            ps.flags = f2u(sm.w);
            local = (int)f2u(sa.w);
            if (!DEFER) {
              done_sum[lane] = sa.x; done_sum[64u + lane] = sa.y; done_sum[128u + lane] = sa.z;
              if (squares) {
                const q4 sb = stash[5u * kStashEntries + e];
                sq_sum[lane] = sb.x; sq_sum[64u + lane] = sb.y; sq_sum[128u + lane] = sb.z;
              }
            }
          }
          sl = decode_slot(p, pixel);
          rc = make_rand_ctx(sl.x, sl.y, 0, p.frame);  // the pixel's noise shift: once per slot
          has_slot = true;
          // nothing to do for this slot (a queue may list slots that have finished): it retires at once
          retiring = !(ps.flags & (kAlive | kFoldPending)) && local >= sl.count;
        }
        grab_pos += take;
        need = __ballot(!has_slot);
      }
    }
    const unsigned long long running = __ballot(has_slot && !retiring);
    if (running == 0ull && __ballot(has_slot) == 0ull) break;  // the queue is empty and every slot is back in memory
    const bool active = has_slot && !retiring;
    ShadowRay sr;
    sr.valid = false;
    if (active) { PT_PROF(PF_BOUNCE); }
    PT_TIC(t_regen);
    if (active && !(ps.flags & kAlive)) {
      PT_PROF(PF_REGEN);
      if (ps.flags & kFoldPending) {  // a sample just ended: fold it into the slot's sum
        done_sum[lane] += ps.sum.x; done_sum[64u + lane] += ps.sum.y; done_sum[128u + lane] += ps.sum.z;
        if (squares) {
          q4 b = q4{sq_sum[lane], sq_sum[64u + lane], sq_sum[128u + lane], 0.0f};
          add_squares(b, ps.sum);
          sq_sum[lane] = b.x; sq_sum[64u + lane] = b.y; sq_sum[128u + lane] = b.z;
        }
        ps.flags &= ~kFoldPending;
      }
      if (local >= sl.count) {
        retiring = true;
      } else {
        generate_path(lds_camera(cam_lds), rc, hash_sample_index(p, sl, local), p.width, p.height, start_seed(p, sl, local, ps.seed), ps);
        local++;
      }
    }
    PT_TOC(t_regen, PT_T_REGEN);
    if (active && !retiring) {
      rc.s = hash_sample_index(p, sl, local - 1);
      PT_TIC(t_trace);
      const Hit h = trace(ps.o, ps.d, 1e16f, false, ps.origin);
      PT_TOC(t_trace, PT_T_TRACE);
      rays_rad++;
      TrainVertex tv;
      PT_TIC(t_shade);
      shade_hit<GUIDED, TRAIN>(sc, p.guide, sp, rc, ps, h, sr, tv);
      if (TRAIN) capture_train(p, tv, ps.seed, pixel, captured);
      PT_TOC(t_shade, PT_T_SHADE);
      if (!DEFER && sr.valid) {
        PT_PROF(PF_SHADOW);
        rays_sh++;
        const Hit o = trace(sr.o, sr.d, sr.tmax, true, sr.origin);
        if (o.prim < 0) ps.sum += sr.contrib;
      }
    }
    if (DEFER) {
      PT_TIC(t_enq);
      const unsigned long long made = __ballot(sr.valid);
      if (sr.valid) {
        const uint32_t rank = (uint32_t)__popcll(made & lanes_below);
        const uint32_t e = (head + waiting + rank) & (kDeferCap - 1u);
        dq[0 * kDeferCap + e] = sr.o.x; dq[1 * kDeferCap + e] = sr.o.y; dq[2 * kDeferCap + e] = sr.o.z;
        dq[3 * kDeferCap + e] = sr.d.x; dq[4 * kDeferCap + e] = sr.d.y; dq[5 * kDeferCap + e] = sr.d.z;
        dq[6 * kDeferCap + e] = sr.tmax;
        dq[7 * kDeferCap + e] = u2f(sr.origin | (lane << kPersistOriginBits));
        dq[8 * kDeferCap + e] = sr.contrib.x; dq[9 * kDeferCap + e] = sr.contrib.y; dq[10 * kDeferCap + e] = sr.contrib.z;
        last_entry[lane] = queued_total + rank + 1u;
      }
      waiting += (uint32_t)__popcll(made);
      queued_total += (uint32_t)__popcll(made);
      PT_TOC(t_enq, PT_T_ENQUEUE);
      // 64 at a time; what is left, at whatever occupancy, once only lanes that wait for it remain
      if (waiting >= 64u || (waiting > 0u && __ballot(has_slot && !retiring) == 0ull)) {
        const uint32_t cnt = waiting < 64u ? waiting : 64u;
        if (lane < cnt) trace_deferred((head + lane) & (kDeferCap - 1u));
        head = (head + cnt) & (kDeferCap - 1u);
        waiting -= cnt;
        traced_total += cnt;
      }
    }
  }
  PT_TOC(t_loop, PT_T_LOOP);
#ifdef PT_PROFILE
  if (lane == 0u) atomicMax(&g_prof_global[59], (unsigned long long)wall_clock64());
  __syncthreads();
  if (threadIdx.x < 64 && g_prof_lds[threadIdx.x]) atomicAdd(&g_prof_global[threadIdx.x], (unsigned long long)g_prof_lds[threadIdx.x]);
#endif
  if (TRAIN) flush_train_count(p, captured);
  for (int off = 32; off > 0; off >>= 1) {
    rays_rad += __shfl_down(rays_rad, off);
    rays_sh += __shfl_down(rays_sh, off);
  }
  if (lane == 0u) {
    if (rays_rad) atomicAdd(&p.ray_counters[0], (unsigned long long)rays_rad);
    if (rays_sh) atomicAdd(&p.ray_counters[1], (unsigned long long)rays_sh);
  }
}

// ------------------------------------------------------------------------- adaptive
// Deterministic form of the reference's per-launch noise test (optix_device.cu:1011-1060):
// every pixel that took part in the round computes its own noise from (sum, sum of squares, n),
// raises the 3x3 neighbourhood of the noise map to it, and -- in a second pass, so that the
// outcome does not depend on scheduling -- stays active while the 3x3 maximum around it
// reaches the threshold.  The map is cleared at active pixels only (cuda_kernels.cu:18-25),
// so a value raised at a pixel that has since stopped persists, as in the reference.
__device__ __forceinline__ float pixel_noise(q4 s1, q4 s2, int n, int ch) {
  const float inv_n = 1.0f / (float)n, inv_n1 = 1.0f / (float)(n - 1);
  const f3 mean = mk3(s1.x * inv_n, s1.y * inv_n, s1.z * inv_n);
  const f3 var = mk3(fmaxf(s2.x - s1.x * mean.x, 0.f) * inv_n1, fmaxf(s2.y - s1.y * mean.y, 0.f) * inv_n1,
                     fmaxf(s2.z - s1.z * mean.z, 0.f) * inv_n1);
  const f3 w = mk3(0.2126f, 0.7152f, 0.0722f);
  const float noise_rgb = sqrtf(fmaxf(dot(var, w), 0.f)) / fmaxf(dot(mean, w), 1e-6f);
  const float mean_b = ch == 0 ? mean.x : (ch == 1 ? mean.y : mean.z);
  const float var_b = ch == 0 ? var.x : (ch == 1 ? var.y : var.z);
  const float noise_bayer = sqrtf(var_b) / fmaxf(mean_b, 1e-6f);
  return fmaxf(noise_rgb, noise_bayer);
}
__global__ void k_noise_clear(const uint32_t* active, uint32_t n, uint32_t* noise_map) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) noise_map[active[i]] = 0u;
}
// statistics of a pixel: its slots added up in chunk order
struct PixelStats {
  q4 s1, s2;
  int n;
};
__device__ __forceinline__ PixelStats pixel_stats(const q4* acc, const q4* acc2, const int* started, uint32_t px,
                                                  uint32_t npx, int chunks) {
  PixelStats r{acc[px], acc2[px], started[px]};
  for (int k = 1; k < chunks; ++k) {
    const size_t i = (size_t)k * npx + px;
    const q4 a = acc[i], b = acc2[i];
    r.s1.x += a.x; r.s1.y += a.y; r.s1.z += a.z;
    r.s2.x += b.x; r.s2.y += b.y; r.s2.z += b.z;
    r.n += started[i];
  }
  return r;
}
__global__ void k_noise_raise(const uint32_t* active, uint32_t n, const q4* acc, const q4* acc2, const int* started,
                              int chunks, uint32_t* noise_map, float* own_noise, int* total, int w, int h, int pattern,
                              int min_samples) {
  const uint32_t npx = (uint32_t)w * (uint32_t)h;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t px = active[i];
    const PixelStats st = pixel_stats(acc, acc2, started, px, npx, chunks);
    const int cnt = st.n;
    float noise = -1.0f;
    if (cnt >= min_samples && cnt > 1) {
      const int y = (int)(px / (uint32_t)w), x = (int)(px - (uint32_t)y * (uint32_t)w);
      noise = pixel_noise(st.s1, st.s2, cnt, bayer_channel_for(x, y, pattern));
      const uint32_t bits = f2u(noise);  // noise >= 0: float order == unsigned order of the bits
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int xx = x + dx, yy = y + dy;
          if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
          atomicMax(&noise_map[(size_t)yy * (size_t)w + (size_t)xx], bits);
        }
    }
    own_noise[px] = noise;
    total[px] = cnt;
  }
}
__global__ void __launch_bounds__(256) k_noise_select(const uint32_t* active, uint32_t n, const uint32_t* noise_map,
                                                      const float* own_noise, const int* total, int w, int h, int spp,
                                                      float threshold, uint32_t* out, uint32_t* out_count) {
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    bool keep = false;
    uint32_t px = 0;
    if (i < n) {
      px = active[i];
      const float own = own_noise[px];
      keep = total[px] < spp;
      if (own >= 0.f) {
        const int y = (int)(px / (uint32_t)w), x = (int)(px - (uint32_t)y * (uint32_t)w);
        float mx = own;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx, yy = y + dy;
            if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
            mx = fmaxf(mx, u2f(noise_map[(size_t)yy * (size_t)w + (size_t)xx]));
          }
        keep = keep && mx >= threshold;
      }
    }
    const uint32_t slot = wave_append_slot(keep, out_count);
    if (keep) out[slot] = px;
  }
}
__global__ void k_expand_active(const uint32_t* active, uint32_t n, int chunks, uint32_t npx, uint32_t* queue, int chunk_major) {
  const uint32_t total = n * (uint32_t)chunks;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t a, c;
    if (chunk_major) { c = i / n; a = i - c * n; }
    else { a = i / (uint32_t)chunks; c = i - a * (uint32_t)chunks; }
    queue[i] = active[a] + c * npx;
  }
}

// Statistics of every pixel as eight floats (sum r g b, samples taken, sum of squares r g b, 0): the form in
// which ranks of a sample-sharded adaptive render add their shares together, and its inverse.
__global__ void k_pack_stats(const q4* acc, const q4* acc2, const int* started, int chunks, float* stats, uint32_t npx) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const PixelStats st = pixel_stats(acc, acc2, started, i, npx, chunks);
    q4* out = reinterpret_cast<q4*>(stats) + 2 * (size_t)i;
    out[0] = q4{st.s1.x, st.s1.y, st.s1.z, (float)st.n};
    out[1] = q4{st.s2.x, st.s2.y, st.s2.z, 0.f};
  }
}
__global__ void k_unpack_stats(const float* stats, q4* acc, q4* acc2, int* count, uint32_t npx) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const q4* in = reinterpret_cast<const q4*>(stats) + 2 * (size_t)i;
    const q4 a = in[0], b = in[1];
    acc[i] = q4{a.x, a.y, a.z, 0.f};
    acc2[i] = q4{b.x, b.y, b.z, 0.f};
    count[i] = (int)(a.w + 0.5f);
  }
}

// In-library sample shards of an adaptive render: add a peer's packed statistics to this shard's, and the mean image
// of the finished render from the totals.
__global__ void k_add_stats(float* own, const float* peer, size_t n) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) own[i] += peer[i];
}
__global__ void k_stats_mean(const float* stats, float* out, uint32_t npx) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const q4 a = reinterpret_cast<const q4*>(stats)[2 * (size_t)i];
    const float inv = a.w > 0.0f ? 1.0f / a.w : 0.0f;
    out[3 * i + 0] = a.x * inv; out[3 * i + 1] = a.y * inv; out[3 * i + 2] = a.z * inv;
  }
}

// ------------------------------------------------------------------------ utilities
__global__ void k_iota(uint32_t* dst, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = i;
}
__global__ void k_finalize(const q4* acc, const int* started, float* out, uint32_t npx, int chunks, int mode) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    q4 a = acc[i];
    int c = started[i];
    for (int k = 1; k < chunks; ++k) {  // fixed order: results do not depend on scheduling
      const q4 b = acc[(size_t)k * npx + i];
      a.x += b.x; a.y += b.y; a.z += b.z;
      c += started[(size_t)k * npx + i];
    }
    if (mode == 0) {
      const float inv = c > 0 ? 1.0f / (float)c : 0.0f;
      out[3 * i + 0] = a.x * inv; out[3 * i + 1] = a.y * inv; out[3 * i + 2] = a.z * inv;
    } else {
      out[3 * i + 0] += a.x; out[3 * i + 1] += a.y; out[3 * i + 2] += a.z;
    }
  }
}
// Sample-sharded renders inside one process: a shard's per-pixel totals as {sum r, g, b, samples taken}, and
// the mean over all shards, added up in shard order so the result does not depend on timing.
__global__ void k_shard_sums(const q4* acc, const int* started, q4* sum4, uint32_t npx, int chunks) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    q4 a = acc[i];
    int c = started[i];
    for (int k = 1; k < chunks; ++k) {
      const q4 b = acc[(size_t)k * npx + i];
      a.x += b.x; a.y += b.y; a.z += b.z;
      c += started[(size_t)k * npx + i];
    }
    sum4[i] = q4{a.x, a.y, a.z, (float)c};
  }
}
__global__ void k_reduce_shards(const q4* own, const q4* peers, int npeers, float* out, uint32_t npx) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    q4 t = own[i];
    for (int k = 0; k < npeers; ++k) {
      const q4 b = peers[(size_t)k * npx + i];
      t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    }
    const float inv = t.w > 0.0f ? 1.0f / t.w : 0.0f;
    out[3 * i + 0] = t.x * inv; out[3 * i + 1] = t.y * inv; out[3 * i + 2] = t.z * inv;
  }
}
__global__ void k_add_counts(const int* started, float* count, uint32_t npx, int chunks) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    int c = 0;
    for (int k = 0; k < chunks; ++k) c += started[(size_t)k * npx + i];
    count[i] += (float)c;
  }
}
// optix_device.cu:944,990-997,1069: the mosaic value is the CFA channel of the pixel's mean, clamped at 0
__global__ void k_mosaic(const float* rgb, float* raw, int w, int h, int pattern, float inv_div) {
  const uint32_t npx = (uint32_t)w * (uint32_t)h;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const int x = (int)(i % (uint32_t)w), y = (int)(i / (uint32_t)w);
    const int ch = bayer_channel_for(x, y, pattern);
    raw[i] = fmaxf(rgb[3 * i + ch] * inv_div, 0.f);
  }
}

static inline size_t stack_bytes(const RenderParams& p, const LaunchCfg& c) {
  return (size_t)c.threads * (size_t)(p.stack_depth > 0 ? p.stack_depth : 1) * sizeof(StackEntry);
}

void launch_generate(const RenderParams& p, int qin, const LaunchCfg& c) {
  k_generate<<<c.blocks, c.threads, 0, c.stream>>>(p, qin);
  k_tile_scan<<<1, 1024, 0, c.stream>>>(p, qin);
  k_tile_gather<<<c.blocks, c.threads, 0, c.stream>>>(p, qin);
  PT_KCHECK();
}
void launch_extend(const RenderParams& p, int q, const LaunchCfg& c) {
  if (p.scene.inst_records) k_extend<true><<<c.blocks, c.threads, stack_bytes(p, c), c.stream>>>(p, q);
  else k_extend<false><<<c.blocks, c.threads, stack_bytes(p, c), c.stream>>>(p, q);
  PT_KCHECK();
}
// GUIDED and TRAIN are compiled in separately: a guided render without a training ring (or the other way round)
// does not carry the other's registers.
template <bool GUIDED, bool TRAIN>
static void launch_shade_t(const RenderParams& p, int q, const LaunchCfg& c) {
  if (p.scene.inst_records) k_shade<GUIDED, TRAIN, true><<<c.blocks, c.threads, 0, c.stream>>>(p, q);
  else k_shade<GUIDED, TRAIN, false><<<c.blocks, c.threads, 0, c.stream>>>(p, q);
}
void launch_shade(const RenderParams& p, int q, const LaunchCfg& c) {
  const bool g = p.guiding != 0, t = p.train != nullptr;
  if (g && t) launch_shade_t<true, true>(p, q, c);
  else if (g) launch_shade_t<true, false>(p, q, c);
  else if (t) launch_shade_t<false, true>(p, q, c);
  else launch_shade_t<false, false>(p, q, c);
  PT_KCHECK();
}
void launch_occlude(const RenderParams& p, int q, const LaunchCfg& c) {
  if (p.scene.inst_records) k_occlude<true><<<c.blocks, c.threads, stack_bytes(p, c), c.stream>>>(p, q);
  else k_occlude<false><<<c.blocks, c.threads, stack_bytes(p, c), c.stream>>>(p, q);
  PT_KCHECK();
}
template <bool GUIDED, bool TRAIN>
static void launch_bounce_t(const RenderParams& p, int qin, int bounces, int compact, QueueCounters cnt, const LaunchCfg& c,
                            bool lds_scene, bool defer, size_t lds) {
  if (defer) {
    if (lds_scene) k_bounce_defer<GUIDED, TRAIN, true><<<c.blocks, c.threads, lds, c.stream>>>(p, qin, bounces, compact, cnt);
    else k_bounce_defer<GUIDED, TRAIN, false><<<c.blocks, c.threads, lds, c.stream>>>(p, qin, bounces, compact, cnt);
  } else {
    if (lds_scene) k_bounce<GUIDED, TRAIN, true><<<c.blocks, c.threads, lds, c.stream>>>(p, qin, bounces, compact, cnt);
    else k_bounce<GUIDED, TRAIN, false><<<c.blocks, c.threads, lds, c.stream>>>(p, qin, bounces, compact, cnt);
  }
}
// A scene is staged in LDS when the block still fits four to a CU with it (160 KB of LDS per CU; the kernels are
// compiled for four waves per SIMD, so a fifth block would find no registers): `other` is what the block needs
// besides.  PT_LDS_SCENE_KB replaces the rule by a limit on the scene's size alone (experiments).
constexpr size_t kLdsBlockBudget = 40960;
static bool scene_goes_to_lds(const RenderParams& p, size_t other, size_t* scene_bytes_out = nullptr) {
  static const long kb = getenv("PT_LDS_SCENE_KB") ? atol(getenv("PT_LDS_SCENE_KB")) : -1;
  const size_t scene_bytes = 16u * (size_t)staged_scene_words(p.scene_nodes, p.scene.num_items, p.scene.num_tris);
  if (scene_bytes_out) *scene_bytes_out = scene_bytes;
  if (!p.scene_in_lds || !p.scene.wide_src) return false;
  return kb >= 0 ? scene_bytes <= (size_t)kb * 1024u : other + scene_bytes <= kLdsBlockBudget;
}
// launch after launch: stacks, camera, and the shadow queues or the slot totals
static size_t bounce_lds_bytes(const RenderParams& p, int threads, bool& lds_scene) {
  const bool defer = p.defer_shadow && !p.acc2;
  const size_t other = (size_t)threads * (size_t)(p.stack_depth > 0 ? p.stack_depth : 1) * sizeof(StackEntry) + kCamLdsBytes +
                       (defer ? (size_t)(threads / 64) * kDeferWords * sizeof(float) : (size_t)threads * kTotalsWords * sizeof(float));
  size_t scene_bytes;
  lds_scene = scene_goes_to_lds(p, other, &scene_bytes);
  return other + (lds_scene ? scene_bytes : 0);
}
void launch_bounce(const RenderParams& p, int qin, int bounces, int compact, QueueCounters cnt, const LaunchCfg& c) {
  bool lds_scene;
  const size_t lds = bounce_lds_bytes(p, c.threads, lds_scene);
  const bool defer = p.defer_shadow && !p.acc2;
  const bool g = p.guiding != 0, t = p.train != nullptr;
  if (g && t) launch_bounce_t<true, true>(p, qin, bounces, compact, cnt, c, lds_scene, defer, lds);
  else if (g) launch_bounce_t<true, false>(p, qin, bounces, compact, cnt, c, lds_scene, defer, lds);
  else if (t) launch_bounce_t<false, true>(p, qin, bounces, compact, cnt, c, lds_scene, defer, lds);
  else launch_bounce_t<false, false>(p, qin, bounces, compact, cnt, c, lds_scene, defer, lds);
  PT_KCHECK();
}
// (fresh_queue is read at run time: a variant compiled for queues that are not fresh ran adaptive renders 4 % slower,
// 349 against 335 ms, and one compiled for fresh queues changed nothing)
template <bool GUIDED, bool TRAIN, int SCENE, bool DEFER>
static void launch_persist_f(const RenderParams& p, int qin, int cnt_in, const LaunchCfg& c, size_t lds) {
  k_bounce_persist<GUIDED, TRAIN, SCENE, DEFER><<<c.blocks, c.threads, lds, c.stream>>>(p, qin, cnt_in);
}
template <bool GUIDED, bool TRAIN>
static void launch_persist_t(const RenderParams& p, int qin, int cnt_in, const LaunchCfg& c, bool lds_scene, bool defer, size_t lds) {
  static const bool one_node_env = !(getenv("PT_ONE_NODE") && atoi(getenv("PT_ONE_NODE")) == 0);
  const bool one_node = lds_scene && p.scene.wide_one_node != 0u && one_node_env;
  if (defer) {
    if (one_node) launch_persist_f<GUIDED, TRAIN, 2, true>(p, qin, cnt_in, c, lds);
    else if (lds_scene) launch_persist_f<GUIDED, TRAIN, 1, true>(p, qin, cnt_in, c, lds);
    else launch_persist_f<GUIDED, TRAIN, 0, true>(p, qin, cnt_in, c, lds);
  } else {
    if (one_node) launch_persist_f<GUIDED, TRAIN, 2, false>(p, qin, cnt_in, c, lds);
    else if (lds_scene) launch_persist_f<GUIDED, TRAIN, 1, false>(p, qin, cnt_in, c, lds);
    else launch_persist_f<GUIDED, TRAIN, 0, false>(p, qin, cnt_in, c, lds);
  }
}
// The deferred form: a shadow ray's origin primitive shares a word with its owner lane, which bounds the triangle
// count; beyond it shadow rays are traced in place.
static bool persist_defers(const RenderParams& p) {
  return p.defer_shadow && !p.acc2 && p.scene.num_tris < (1u << kPersistOriginBits) - 1u;
}
// (every render the fused kernels take: kept as the one place to say otherwise)
bool bounce_persist_usable(const RenderParams&) { return true; }
static size_t persist_lds_bytes(const RenderParams& p, int threads, bool& lds_scene) {
  const size_t other = (size_t)threads * (size_t)(p.stack_depth > 0 ? p.stack_depth : 1) * sizeof(StackEntry) +
                       (size_t)(threads / 64) * persist_words(persist_defers(p)) * sizeof(float) + kCamLdsBytes +
                       (p.fresh_queue ? 0u : (size_t)(threads / 64) * kStashWords * sizeof(float));
  size_t scene_bytes;
  lds_scene = scene_goes_to_lds(p, other, &scene_bytes);
  return other + (lds_scene ? scene_bytes : 0);
}
// Would a fixed-count render of this scene by the one-launch kernel, shadow rays traced in place, have the scene in LDS?
bool persist_in_place_stages_scene(const RenderParams& p, int threads) {
  RenderParams q = p;
  q.defer_shadow = 0;
  q.fresh_queue = 1;
  bool lds_scene;
  persist_lds_bytes(q, threads, lds_scene);
  return lds_scene;
}
// The whole of queue qin (its length in counters[cnt_in]) in one launch; counters[CNT_GRAB] must be zero.
void launch_bounce_persist(const RenderParams& p, int qin, int cnt_in, const LaunchCfg& c) {
  bool lds_scene;
  const size_t lds = persist_lds_bytes(p, c.threads, lds_scene);
  const bool g = p.guiding != 0, t = p.train != nullptr;
  const bool defer = persist_defers(p);
  if (g && t) launch_persist_t<true, true>(p, qin, cnt_in, c, lds_scene, defer, lds);
  else if (g) launch_persist_t<true, false>(p, qin, cnt_in, c, lds_scene, defer, lds);
  else if (t) launch_persist_t<false, true>(p, qin, cnt_in, c, lds_scene, defer, lds);
  else launch_persist_t<false, false>(p, qin, cnt_in, c, lds_scene, defer, lds);
  PT_KCHECK();
}
// Resident blocks per CU of the fused kernel launch_bounce would pick for these parameters, as the runtime counts
// them, and the dynamic LDS bytes of that launch.
int bounce_occupancy(const RenderParams& p, int threads, size_t* lds_out, bool one_launch) {
  if (one_launch && bounce_persist_usable(p)) {
    bool lds_scene;
    const size_t lds = persist_lds_bytes(p, threads, lds_scene);
    if (lds_out) *lds_out = lds;
    int n = 0;
    hipError_t e;
    if (persist_defers(p))
      e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce_persist<false, false, 1, true>, threads, lds)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce_persist<false, false, 0, true>, threads, lds);
    else
      e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce_persist<false, false, 1, false>, threads, lds)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce_persist<false, false, 0, false>, threads, lds);
    return e == hipSuccess ? n : -1;
  }
  bool lds_scene;
  const size_t lds = bounce_lds_bytes(p, threads, lds_scene);
  const bool defer = p.defer_shadow && !p.acc2;
  if (lds_out) *lds_out = lds;
  int n = 0;
  const hipError_t e = defer ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce_defer<false, false, true>, threads, lds)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce<false, false, true>, threads, lds);
  return e == hipSuccess ? n : -1;
}
// Profiling build only: copy out (and clear) the section counters; returns 0 in the shipped build.
int read_profile_counters(unsigned long long* out64) {
#ifdef PT_PROFILE
  unsigned long long zero[64] = {0};
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof_global), sizeof(zero)) != hipSuccess) return 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof_global), zero, sizeof(zero)) != hipSuccess) return 0;
  return 32;  // 18 section pairs, then PT_T_COUNT time sums from word 40
#else
  (void)out64;
  return 0;
#endif
}
void launch_iota(uint32_t* dst, uint32_t n, hipStream_t s) {
  k_iota<<<1024, 256, 0, s>>>(dst, n);
  PT_KCHECK();
}
void launch_pack_stats(const q4* acc, const q4* acc2, const int* started, int chunks, float* stats, uint32_t npx, hipStream_t s) {
  k_pack_stats<<<1024, 256, 0, s>>>(acc, acc2, started, chunks, stats, npx);
  PT_KCHECK();
}
void launch_unpack_stats(const float* stats, q4* acc, q4* acc2, int* count, uint32_t npx, hipStream_t s) {
  k_unpack_stats<<<1024, 256, 0, s>>>(stats, acc, acc2, count, npx);
  PT_KCHECK();
}
void launch_add_stats(float* own, const float* peer, size_t n, hipStream_t s) {
  k_add_stats<<<1024, 256, 0, s>>>(own, peer, n);
  PT_KCHECK();
}
void launch_stats_mean(const float* stats, float* out_rgb, uint32_t npx, hipStream_t s) {
  k_stats_mean<<<1024, 256, 0, s>>>(stats, out_rgb, npx);
  PT_KCHECK();
}
void launch_noise_round(const AdaptiveBuffers& a, int which, uint32_t n_active, const q4* acc, const q4* acc2,
                        const int* started, int chunks, int w, int h, int pattern, int min_samples, int spp,
                        float threshold, hipStream_t s) {
  const int blocks = (int)((n_active + 255u) / 256u < 2048u ? (n_active + 255u) / 256u : 2048u);
  if (blocks <= 0) return;
  k_noise_clear<<<blocks, 256, 0, s>>>(a.active[which], n_active, a.noise_map);
  PT_KCHECK();
  k_noise_raise<<<blocks, 256, 0, s>>>(a.active[which], n_active, acc, acc2, started, chunks, a.noise_map, a.own_noise,
                                       a.total, w, h, pattern, min_samples);
  PT_KCHECK();
  k_noise_select<<<blocks, 256, 0, s>>>(a.active[which], n_active, a.noise_map, a.own_noise, a.total, w, h, spp, threshold,
                                        a.active[which ^ 1], a.count);
  PT_KCHECK();
}
void launch_expand_active(const uint32_t* active, uint32_t n_active, int chunks, uint32_t npx, uint32_t* queue, hipStream_t s) {
  if (n_active == 0) return;
  // chunk by chunk (all active pixels' first slots, then their second ...): neighbours in the queue are then neighbours
  // in the state arrays; pixel by pixel (a pixel's slots side by side) measured 4 % slower at 1080p
  static const int chunk_major = getenv("PT_EXPAND_ORDER") ? atoi(getenv("PT_EXPAND_ORDER")) : 1;
  k_expand_active<<<1024, 256, 0, s>>>(active, n_active, chunks, npx, queue, chunk_major);
  PT_KCHECK();
}
void launch_finalize(const q4* acc, const int* started, float* out_rgb, uint32_t npx, int chunks, int mode, hipStream_t s) {
  k_finalize<<<1024, 256, 0, s>>>(acc, started, out_rgb, npx, chunks, mode);
  PT_KCHECK();
}
void launch_shard_sums(const q4* acc, const int* started, q4* sum4, uint32_t npx, int chunks, hipStream_t s) {
  k_shard_sums<<<1024, 256, 0, s>>>(acc, started, sum4, npx, chunks);
  PT_KCHECK();
}
void launch_reduce_shards(const q4* own, const q4* peers, int npeers, float* out_rgb, uint32_t npx, hipStream_t s) {
  k_reduce_shards<<<1024, 256, 0, s>>>(own, peers, npeers, out_rgb, npx);
  PT_KCHECK();
}
void launch_add_counts(const int* started, float* count, uint32_t npx, int chunks, hipStream_t s) {
  k_add_counts<<<1024, 256, 0, s>>>(started, count, npx, chunks);
  PT_KCHECK();
}
void launch_mosaic(const float* rgb, float* raw, int w, int h, int pattern, float divisor, hipStream_t s) {
  k_mosaic<<<1024, 256, 0, s>>>(rgb, raw, w, h, pattern, 1.0f / divisor);
  PT_KCHECK();
}

// ----------------------------------------------------------------- stage-level debug
__global__ void k_dbg_pcg(unsigned long long seed, int n, uint32_t* out_u, float* out_f) {
  if (blockIdx.x || threadIdx.x) return;
  uint64_t s = seed, s2 = seed;
  for (int i = 0; i < n; ++i) {
    out_u[i] = pcg(s);
    out_f[i] = rnd(s2);
  }
}
__global__ void k_dbg_blue_noise(int n, const int* xysf, float* uv, float* jitter) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float u, v;
  blue_noise(xysf[4 * i], xysf[4 * i + 1], xysf[4 * i + 2], xysf[4 * i + 3], u, v);
  uv[2 * i] = u; uv[2 * i + 1] = v;
  jitter[2 * i] = sample_tri(u); jitter[2 * i + 1] = sample_tri(v);
}
__global__ void k_dbg_camera(CameraView cam, int w, int h, int n, const int* xy, const float* jitter, float* dirs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 d = camera_dir(cam, xy[2 * i], xy[2 * i + 1], jitter[2 * i], jitter[2 * i + 1], w, h);
  dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
}
__global__ void __launch_bounds__(256) k_trace_rays(SceneView sc, int stack_depth, int n, const float* o, const float* d,
                                                    const float* tmin, const float* tmax, int any_hit, int* prim,
                                                    float* t, float* b1, float* b2, int* inst) {
  LdsStack st = lds_stack();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Hit h;
    if (sc.inst_records) {  // instanced scenes are walked step by step, as the staged kernels walk them
      TraceStateT<true> ts;
      ts.begin(sc, ro, rd, tmax[i], 0u);
      while (!ts.done()) {
        if (ts.wants_node()) ts.node_step(sc, ro, rd, tmin[i], st, stack_depth);
        if (ts.wants_tri() && ts.tri_step(sc, ro, rd, tmin[i], any_hit != 0, st, stack_depth)) ts.finish();
      }
      h = ts.hit;
      if (inst) inst[i] = (int)ts.hit_inst;
    } else {
      h = trace_ray(sc, ro, rd, tmin[i], tmax[i], any_hit != 0, st, stack_depth);
      if (inst) inst[i] = -1;
    }
    if (any_hit) {
      prim[i] = h.prim >= 0 ? 1 : 0;
    } else {
      prim[i] = h.prim;
      float r1 = h.b1, r2 = h.b2;
      if (h.prim >= 0) to_input_order(sc.shade[h.prim].corner_order, r1, r2);
      t[i] = h.t; b1[i] = r1; b2[i] = r2;
    }
  }
}
__global__ void __launch_bounds__(256) k_dbg_shade(SceneView sc, GuideView gv, int guiding, int stack_depth, int n,
                                                   DbgPathRec* recs, const int* prim, const float* b1, const float* b2,
                                                   int frame, int max_depth, int* shadow_traced) {
  LdsStack st = lds_stack();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    DbgPathRec r = recs[i];
    PathState ps;
    ps.o = mk3(r.origin[0], r.origin[1], r.origin[2]);
    ps.d = mk3(r.direction[0], r.direction[1], r.direction[2]);
    ps.T = mk3(r.throughput[0], r.throughput[1], r.throughput[2]);
    ps.sum = mk3(0.f);
    ps.seed = r.seed;
    ps.prev_pdf = r.prev_pdf_bsdf;
    const uint32_t dim = r.rng_dim < 2 ? (uint32_t)r.rng_dim : 2u;
    ps.flags = ((uint32_t)r.depth & kDepthMask) | (dim << kDimShift) | (r.prev_pdf_valid ? kPrevValid : 0u) | kAlive;
    const RandCtx rc = make_rand_ctx(r.px, r.py, r.sample, frame);
    Hit h{prim[i], 0.f, b1[i], b2[i]};  // barycentrics arrive in the order the triangle was given
    if (h.prim >= 0) to_record_order(sc.shade[h.prim].corner_order, h.b1, h.b2);
    ShadowRay sr;
    const ShadeParams sp{frame, max_depth, guiding};
    TrainVertex tv;
    shade_hit<true, false>(sc, gv, sp, rc, ps, h, sr, tv);
    if (sr.valid) {
      const Hit o = trace_ray(sc, sr.o, sr.d, 0.0f, sr.tmax, true, st, stack_depth);
      if (o.prim < 0) ps.sum += sr.contrib;
    }
    shadow_traced[i] = sr.valid ? 1 : 0;
    // radiance slot returns the bounce's contribution radiance * throughput
    r.radiance[0] = ps.sum.x; r.radiance[1] = ps.sum.y; r.radiance[2] = ps.sum.z;
    r.throughput[0] = ps.T.x; r.throughput[1] = ps.T.y; r.throughput[2] = ps.T.z;
    r.origin[0] = ps.o.x; r.origin[1] = ps.o.y; r.origin[2] = ps.o.z;
    r.direction[0] = ps.d.x; r.direction[1] = ps.d.y; r.direction[2] = ps.d.z;
    r.seed = ps.seed;
    r.depth = (int)ps_depth(ps);
    r.done = (ps.flags & kAlive) ? 0 : 1;
    r.prev_pdf_bsdf = ps.prev_pdf;
    r.prev_pdf_valid = (ps.flags & kPrevValid) ? 1 : 0;
    r.rng_dim = (int)ps_dim(ps);
    recs[i] = r;
  }
}
__global__ void k_dbg_guiding(GuideView gv, int n, const float* pos, const float* u3, const float* wi, int* region,
                              int* lobe, float* dir, float* pdf_dir, float* pdf_wi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 P = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
  const f3 w = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
  const int rid = guiding_region_id(gv, P);
  region[i] = rid;
  lobe[i] = -1;
  dir[3 * i] = dir[3 * i + 1] = dir[3 * i + 2] = 0.f;
  pdf_dir[i] = pdf_wi[i] = 0.f;
  if (rid < 0) return;
  const GuideRegionDev R = gv.regions[rid];
  if (R.lobe_num == 0) return;
  const GuideLobeDev* L = gv.lobes + R.lobe_ofs;
  const int li = guiding_choose_lobe(L, R.lobe_num, R.sum, u3[3 * i]);
  lobe[i] = li;
  pdf_wi[i] = guiding_mixture_pdf(L, R.lobe_num, R.sum, w);
  if (li < 0) return;
  const f3 g = vmf_sample(L[li], u3[3 * i + 1], u3[3 * i + 2]);
  dir[3 * i] = g.x; dir[3 * i + 1] = g.y; dir[3 * i + 2] = g.z;
  pdf_dir[i] = guiding_mixture_pdf(L, R.lobe_num, R.sum, g);
}

void launch_dbg_pcg(unsigned long long seed, int n, uint32_t* out_u, float* out_f, hipStream_t s) {
  k_dbg_pcg<<<1, 1, 0, s>>>(seed, n, out_u, out_f);
  PT_KCHECK();
}
void launch_dbg_blue_noise(int n, const int* xysf, float* uv, float* jitter, hipStream_t s) {
  k_dbg_blue_noise<<<(n + 255) / 256, 256, 0, s>>>(n, xysf, uv, jitter);
  PT_KCHECK();
}
void launch_dbg_camera(CameraView cam, int w, int h, int n, const int* xy, const float* jitter, float* dirs, hipStream_t s) {
  k_dbg_camera<<<(n + 255) / 256, 256, 0, s>>>(cam, w, h, n, xy, jitter, dirs);
  PT_KCHECK();
}
static int blocks_for(int n) {
  int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}
void launch_trace_rays(SceneView sc, int stack_depth, int n, const float* o, const float* d, const float* tmin,
                       const float* tmax, int any_hit, int* prim, float* t, float* b1, float* b2, int* inst, hipStream_t s) {
  const size_t lds = 256 * (size_t)(stack_depth > 0 ? stack_depth : 1) * sizeof(StackEntry);
  k_trace_rays<<<blocks_for(n), 256, lds, s>>>(sc, stack_depth, n, o, d, tmin, tmax, any_hit, prim, t, b1, b2, inst);
  PT_KCHECK();
}
void launch_dbg_shade(SceneView sc, GuideView gv, int guiding, int stack_depth, int n, DbgPathRec* recs, const int* prim,
                      const float* b1, const float* b2, int frame, int max_depth, int* shadow_traced, hipStream_t s) {
  const size_t lds = 256 * (size_t)(stack_depth > 0 ? stack_depth : 1) * sizeof(StackEntry);
  k_dbg_shade<<<blocks_for(n), 256, lds, s>>>(sc, gv, guiding, stack_depth, n, recs, prim, b1, b2, frame, max_depth,
                                              shadow_traced);
  PT_KCHECK();
}
void launch_dbg_guiding(GuideView gv, int n, const float* pos, const float* u3, const float* wi, int* region, int* lobe,
                        float* dir, float* pdf_dir, float* pdf_wi, hipStream_t s) {
  k_dbg_guiding<<<(n + 255) / 256, 256, 0, s>>>(gv, n, pos, u3, wi, region, lobe, dir, pdf_dir, pdf_wi);
  PT_KCHECK();
}

}  // namespace pt

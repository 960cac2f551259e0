// pt_shade.h -- path state and the shading step of the integrator (a6, a8, a9).
//
// One call of shade_hit() is one execution of the reference's closest-hit program
// (optix_device.cu:400-673) or radiance-miss program (370-375) followed by raygen's
// "accumulate radiance * throughput" (988-997): the contribution of a bounce is weighted
// with the throughput AFTER the bounce's BSDF sampling and Russian roulette.
#pragma once
#include "pt_bvh.h"
#include "pt_guiding.h"
#include "pt_rng.h"
#include "pt_types.h"

#ifndef PT_GUIDE_EARLY_BATCH
#define PT_GUIDE_EARLY_BATCH 1
#endif

#ifndef PT_SHARED_DISK
#define PT_SHARED_DISK 1
#endif

namespace pt {

// Register image of one path.  In HBM it is four 16-byte SoA streams:
//   {o.xyz, T.x} {d.xyz, T.y} {sum.xyz, T.z} {seed_lo, seed_hi, prev_pdf, flags}
struct PathState {
  f3 o, d, T, sum;
  uint64_t seed;
  float prev_pdf;
  uint32_t flags;
  // 1 + primitive the current ray leaves (see trace_ray), 0 if unknown.  Not part of the stored state:
  // it only lets traversal skip tests that cannot succeed.
  uint32_t origin;
};
constexpr uint32_t kDepthMask = 0xffu;
constexpr uint32_t kDimShift = 8u, kDimMask = 3u << 8;
constexpr uint32_t kPrevValid = 1u << 10;
constexpr uint32_t kAlive = 1u << 11;
constexpr uint32_t kFoldPending = 1u << 12;  // the sample's sum has not been added to the pixel yet

PT_HD uint32_t ps_depth(const PathState& p) { return p.flags & kDepthMask; }
PT_HD uint32_t ps_dim(const PathState& p) { return (p.flags & kDimMask) >> kDimShift; }

struct ShadeParams {
  int frame;
  int max_depth;
  int guiding;  // 0/1: GuideView valid
};

// The next-event shadow ray a shading step wants traced; contrib = Lo * T (already weighted),
// added to the path's sample sum when the ray is unoccluded.
struct ShadowRay {
  f3 o, d;
  float tmax;
  f3 contrib;
  bool valid;
  uint32_t origin;  // 1 + primitive it leaves (always known: NEE needs cosS > 0)
};

// What next_rand needs to know about the path's pixel and sample: the sample index that feeds the R2
// sequence and the pixel's Cranley-Patterson shift (pixel_shift).
struct RandCtx {
  int px, py, s;
  float rx, ry;
};
PT_HD RandCtx make_rand_ctx(int px, int py, int s, int frame) {
  RandCtx rc;
  rc.px = px; rc.py = py; rc.s = s;
  pixel_shift(px, py, frame, rc.rx, rc.ry);
  return rc;
}

// optix_device.cu:198-209: the first two dimensions of a path reuse the blue-noise pair
// that jittered the pixel; every later draw is PCG.
PT_HD float next_rand(PathState& ps, const RandCtx& rc) {
  const uint32_t dim = ps_dim(ps);
  if (dim < 2u) {
    float u, v;
    blue_noise_shifted(rc.rx, rc.ry, rc.s, u, v);
    ps.flags += 1u << kDimShift;
    return dim == 0u ? u : v;
  }
  return rnd(ps.seed);
}

// optix_device.cu:275-278 with F0 grey: returns the scalar Schlick term
PT_HD float schlick(float cosTheta, float f0) { return f0 + (1.0f - f0) * pow5(fmaxf(0.0f, 1.0f - cosTheta)); }

PT_HD f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(i, n)); }

// optix_device.cu:285-332
// renormalising a vector that is unit by construction (a unit vector in an orthonormal frame, the mirror image of
// one) changes its last bit at most: left out
PT_HD f3 unit_by_construction(f3 v) { return v; }
// g1v: the Smith term of the view direction, which the caller has already (it depends on N.V and alpha only).
// (T, B): the frame make_onb gives for N; (t1, t2): the point of the unit disk the two random numbers map to,
// sqrt(ux) * (cos, sin)(2 pi uy) -- the caller's diffuse lobe needs the same frame and the same kind of point,
// so both are made once, before the lobes part ways.
PT_HD void sample_ggx_vndf_disk(float t1, float t2, float alpha, f3 N, f3 T, f3 B, f3 V, f3& L, f3& H, float& pdf, float g1v) {
  const f3 Vlocal = mk3(dot(V, T), dot(V, B), dot(V, N));
  const f3 Vh = normalize_chk(mk3(alpha * Vlocal.x, alpha * Vlocal.y, Vlocal.z));
  const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
  const f3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * rsqrt(lensq) : mk3(1.0f, 0.0f, 0.0f);
  const f3 T2 = cross(Vh, T1);
  const float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * fsqrt(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
  const f3 Nh = T1 * t1 + T2 * t2 + Vh * fsqrt(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
  const f3 Hlocal = normalize_chk(mk3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
  H = unit_by_construction(T * Hlocal.x + B * Hlocal.y + N * Hlocal.z);
  L = unit_by_construction(reflect(-V, H));
  const float NoH = fmaxf(dot(N, H), 0.0f);
  const float NoV = fmaxf(dot(N, V), 0.0f);
  const float VoH = fmaxf(dot(V, H), 0.0f);
  const float alpha2 = alpha * alpha;
  const float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
  const float D = alpha2 * rcp(kPi * denom * denom);
  const float G1V = g1v >= 0.0f ? g1v : 2.0f * NoV * rcp(NoV + fsqrt(alpha2 + (1.0f - alpha2) * NoV * NoV));
  pdf = (D * NoH * G1V) * rcp(4.0f * fmaxf(VoH, 1e-6f));
}
PT_HD void sample_ggx_vndf(float ux, float uy, float alpha, f3 N, f3 V, f3& L, f3& H, float& pdf, float g1v = -1.0f) {
  f3 T, B;
  make_onb(N, T, B);
  const float r = fsqrt(ux);
  sample_ggx_vndf_disk(r * cos2pi(uy), r * sin2pi(uy), alpha, N, T, B, V, L, H, pdf, g1v);
}

// optix_device.cu:660-672 (== 468-480).  Returns false when the path ends here.
PT_HD bool depth_and_roulette(PathState& ps, const RandCtx& rc, int max_depth) {
  ps.flags += 1u;  // depth++
  const int depth = (int)ps_depth(ps);
  if (depth >= max_depth) return false;
  if (depth >= max_depth - 2) {
    PT_PROF(PF_RR);
    const float p = max3(ps.T);
    if (next_rand(ps, rc) > p) return false;
    ps.T = ps.T * rcp(fmaxf(p, 1e-3f));
  }
  return true;
}

// What a shading step reports for the guiding trainer (optix_device.cu:431-441): a record is
// captured at every non-delta vertex, with the throughput the path arrived with.
struct TrainVertex {
  bool valid;
  f3 P, dir_in, T_in;
};

// The point a path has arrived at (optix_device.cu:404-429): position, geometric normal and material.
// origin_hint: 1 + the primitive, for the origin-plane skip of the rays that leave it (0: no hint).
struct Surface {
  f3 P, Ng, Base_colour, Emission;
  float Roughness, IOR, Transmission;
  uint32_t origin_hint;
};
PT_HD Surface load_surface(const SceneView& sc, const Hit& hit) {
  const q4* rec = reinterpret_cast<const q4*>(sc.shade + hit.prim);
  const q4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5];
  const f3 v0 = ldv(r0), v1 = ldv(r1), v2 = ldv(r2);
  const float b1 = hit.b1, b2 = hit.b2, b0 = 1.0f - b1 - b2;
  Surface sf;
  sf.P = v0 * b0 + v1 * b1 + v2 * b2;
  sf.Ng = ldv(r3);
  sf.Base_colour = ldv(r4);
  sf.Emission = ldv(r5);
  sf.Roughness = r0.w;
  sf.IOR = r1.w;
  sf.Transmission = r2.w;
  sf.origin_hint = (uint32_t)hit.prim + 1u;
  return sf;
}
// The same for a hit inside an instance (records: 6 q4 per instance, rows of the world -> object matrix, then an
// optional material).  The mesh's record is in object space: the point is taken on the world ray, the normal goes
// through the transpose of world -> object, and a ray that leaves gets no origin hint (the hint names a node of the
// mesh's tree, which other instances share).
PT_HD Surface load_surface_instanced(const SceneView& sc, const Hit& hit, uint32_t inst, f3 o, f3 d) {
  Surface sf = load_surface(sc, hit);
  const q4* x = sc.inst_records + 6u * inst;
  const q4 w0 = x[0], w1 = x[1], w2 = x[2], m0 = x[3], m1 = x[4], m2 = x[5];
  sf.P = o + d * hit.t;
  const f3 n = sf.Ng;
  sf.Ng = normalize3(mk3(w0.x * n.x + w1.x * n.y + w2.x * n.z, w0.y * n.x + w1.y * n.y + w2.y * n.z,
                         w0.z * n.x + w1.z * n.y + w2.z * n.z));
  if (m2.y != 0.0f) {  // the instance brings its own material
    sf.Base_colour = mk3(m0.x, m0.y, m0.z); sf.Roughness = m0.w;
    sf.Emission = mk3(m1.x, m1.y, m1.z); sf.IOR = m1.w;
    sf.Transmission = m2.x;
  }
  sf.origin_hint = 0u;
  return sf;
}

template <bool GUIDED, bool TRAIN>
PT_HD void shade_surface(const SceneView& sc, const GuideView& gv, const ShadeParams& sp, const RandCtx& rc,
                         PathState& ps, const Surface& sf, ShadowRay& sr, TrainVertex& tv) {
  PT_PROF(PF_SHADE);
  const f3 P = sf.P, Ng = sf.Ng, Base_colour = sf.Base_colour, Emission = sf.Emission;
  const float Roughness = sf.Roughness, IOR = sf.IOR, Transmission = sf.Transmission;
  const f3 dir = ps.d;

  const bool is_delta = (Transmission > 0.0f) || (Roughness <= 0.0f);
  if (TRAIN && !is_delta) {
    tv.valid = true;
    tv.P = P;
    tv.dir_in = -dir;
    tv.T_in = ps.T;
  }

  // Both kinds of surface start the same way: the reflectance at normal incidence, and one random number (the
  // dielectric's choice between reflection and refraction, the first coordinate of the opaque surface's light
  // sample).  Done once, ahead of the branch: the few lanes of a wave on glass then run that much less alone.
  float f0 = (IOR - 1.0f) * rcp(IOR + 1.0f);
  f0 = f0 * f0;
  const float u_first = next_rand(ps, rc);

  f3 radiance;
  if (Transmission > 0.0f) {
    PT_PROF(PF_GLASS);
    // optix_device.cu:443-482: smooth dielectric, Schlick Fresnel
    const bool entering = dot(dir, Ng) < 0.0f;
    const f3 N = entering ? Ng : -Ng;
    const float eta = entering ? rcp(IOR) : IOR;
    const float cosi = fmaxf(dot(-N, dir), 0.0f);
    const float Fr = f0 + (1.0f - f0) * pow5(1.0f - cosi);
    bool do_reflect = u_first < Fr;
    f3 Tdir = mk3(0.f);
    if (!do_reflect) {
      // optix_device.cu:266-273
      const float ci = dot(-N, dir);
      const float k = 1.0f - eta * eta * (1.0f - ci * ci);
      if (k < 0.0f) {
        do_reflect = true;
      } else {
        Tdir = dir * eta + N * (eta * ci - fsqrt(k));
      }
    }
    if (do_reflect) {
      ps.o = P + N * 1e-3f;
      ps.d = reflect(dir, N);
    } else {
      ps.o = P - N * 1e-3f;
      ps.d = normalize3(Tdir);
      ps.T = ps.T * Base_colour;
    }
    radiance = Emission;
    ps.flags &= ~kPrevValid;
    ps.prev_pdf = 0.0f;
  } else {
    PT_PROF(PF_OPAQUE);
    // optix_device.cu:484-492
    const f3 V = -dir;
    const float NoV = fmaxf(dot(Ng, V), 0.0f);
    const float F = schlick(NoV, f0);
    const float spec_prob = fminf(0.99f, fmaxf(0.01f, (F + F + F) * (1.0f / 3.0f)));
    const float diff_prob = 1.0f - spec_prob;
    const f3 kd = Base_colour * (1.0f - F);
    const float alpha = fmaxf(Roughness * Roughness, 1e-4f);
    const float alpha2 = alpha * alpha;
    // Smith term of the view direction: the same number in next-event estimation, in the sampler's pdf and in the
    // specular weight (optix_device.cu:531, 328, 577), computed once
    const float G1V_shared = 2.0f * NoV * rcp(NoV + fsqrt(alpha2 + (1.0f - alpha2) * NoV * NoV));
    const float area = 4.0f * sc.light_half[0] * sc.light_half[1];
    const f3 light_n = ld3(sc.light_normal);

    // The guiding field's cell record is asked for here, well before the diffuse branch needs it, so that its
    // trip to memory runs beside the next-event arithmetic (optix_device.cu:596-612 does the lookup in place).
    GuideRegionDev guide_cell{0u, 0u, 0.0f, 0.0f};
    GuideBatch gb;
    if (GUIDED && sp.guiding) {
      guide_cell = guiding_cell(gv, P);
      if (PT_GUIDE_EARLY_BATCH && guide_cell.lobe_num > 0u && guide_cell.lobe_num <= kGuideBatch)
        guiding_load_batch(gv.lobes + guide_cell.lobe_ofs, guide_cell.lobe_num, gb);
    }

    // optix_device.cu:495-548: next-event estimation towards the rectangle light
    f3 Lo = mk3(0.0f);
    {
      const float u = (u_first * 2.0f - 1.0f) * sc.light_half[0];
      const float v = (next_rand(ps, rc) * 2.0f - 1.0f) * sc.light_half[1];
      const f3 lp = mk3(sc.light_pos[0] + u, sc.light_pos[1] + v, sc.light_pos[2]);
      const f3 L = lp - P;
      const float dist2 = fmaxf(dot(L, L), 1e-6f);
      const float dist = fsqrt(dist2);
      const f3 wi = L * rcp(dist);
      const float cosS = fmaxf(0.0f, dot(Ng, wi));
      const float cosL = fmaxf(0.0f, dot(light_n, -wi));
      if (cosS > 0.0f && cosL > 0.0f) {
        PT_PROF(PF_NEE);
        const f3 H = normalize3(V + wi);
        const float VoH = fmaxf(dot(V, H), 0.0f);
        const float NoH = fmaxf(dot(Ng, H), 0.0f);
        const float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
        const float D = alpha2 * rcp(kPi * denom * denom);
        const float G1V = G1V_shared;
        const float G1L = 2.0f * cosS * rcp(cosS + fsqrt(alpha2 + (1.0f - alpha2) * cosS * cosS));
        const float Fh = schlick(VoH, f0);
        const float f_spec = Fh * (D * G1V * G1L * rcp(fmaxf(4.0f * NoV * cosS, 1e-6f)));
        const float pdf_spec = D * NoH * G1V * rcp(fmaxf(4.0f * VoH, 1e-6f)) * spec_prob;
        const float pdf_diff = cosS * kInvPi * diff_prob;
        const float pdf_bsdf = pdf_spec + pdf_diff;
        const float pdf_area = rcp(area);
        const float pdf_light = pdf_area * dist2 * rcp(cosL);
        const float w = pdf_light * rcp(pdf_light + pdf_bsdf);
        const f3 f = kd * kInvPi + mk3(f_spec);
        const f3 contrib = ld3(sc.light_emit) * f * (cosS * cosL * rcp(dist2)) * area;
        Lo = contrib * w;
        sr.valid = true;
        sr.origin = sf.origin_hint;
        sr.o = P + Ng * 1e-3f;
        sr.d = wi;
        sr.tmax = dist - 1e-3f;
      }
    }

    // optix_device.cu:551-561: emitted light, MIS-weighted against the previous bounce's pdf
    f3 emission = Emission;
    if ((ps.flags & kPrevValid) && (Emission.x > 0.0f || Emission.y > 0.0f || Emission.z > 0.0f)) {
      PT_PROF(PF_EMIS_MIS);
      const f3 L = P - ps.o;
      const float dist2 = fmaxf(dot(L, L), 1e-6f);
      const float cosL = fmaxf(0.0f, dot(light_n, -dir));
      const float pdf_light = dist2 * rcp(cosL * area);
      const float w_bsdf = ps.prev_pdf * rcp(ps.prev_pdf + pdf_light);
      emission = emission * w_bsdf;
    }
    radiance = emission;

    const float choose = next_rand(ps, rc);
    const bool take_spec = choose < spec_prob;
    // Either lobe draws two more numbers, maps them to a point of the unit disk -- radius sqrt(one), angle 2 pi
    // times the other: the specular lobe's first number is its radius, the diffuse lobe's first its angle -- and
    // needs make_onb's frame of the normal.  Without a guiding field (whose diffuse lobe draws differently) this is
    // done here for both, so the few lanes of a wave that take the specular lobe run that much less on their own.
    const bool shared_disk = PT_SHARED_DISK && !(GUIDED && sp.guiding);
    float disk_x = 0.0f, disk_y = 0.0f, disk_r2 = 0.0f;
    f3 onb_t = mk3(0.0f), onb_b = mk3(0.0f);
    if (shared_disk) {
      const float ua = next_rand(ps, rc);
      const float ub = next_rand(ps, rc);
      disk_r2 = take_spec ? ua : ub;
      const float ang = take_spec ? ub : ua;
      const float r = fsqrt(disk_r2);
      disk_x = r * cos2pi(ang);
      disk_y = r * sin2pi(ang);
      make_onb(Ng, onb_t, onb_b);
    }
    if (take_spec) {
      PT_PROF(PF_SPEC);
      // optix_device.cu:564-589
      f3 newDir, H;
      float pdf;
      if (shared_disk) {
        sample_ggx_vndf_disk(disk_x, disk_y, alpha, Ng, onb_t, onb_b, V, newDir, H, pdf, G1V_shared);
      } else {
        const float ux = next_rand(ps, rc);
        const float uy = next_rand(ps, rc);
        sample_ggx_vndf(ux, uy, alpha, Ng, V, newDir, H, pdf, G1V_shared);
      }
      const float NoL = fmaxf(dot(Ng, newDir), 0.0f);
      const float VoH = fmaxf(dot(V, H), 0.0f);
      const float NoH = fmaxf(dot(Ng, H), 0.0f);
      const float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
      const float D = alpha2 * rcp(kPi * denom * denom);
      const float G1V = G1V_shared;
      const float G1L = 2.0f * NoL * rcp(NoL + fsqrt(alpha2 + (1.0f - alpha2) * NoL * NoL));
      const float Fh = schlick(VoH, f0);
      const float spec_brdf = Fh * (D * G1V * G1L * rcp(fmaxf(4.0f * NoV * NoL, 1e-6f)));
      ps.o = P + Ng * 1e-3f;
      ps.d = newDir;
      ps.T = ps.T * (spec_brdf * NoL * rcp(fmaxf(pdf * spec_prob, 1e-6f)));
      ps.prev_pdf = pdf * spec_prob;
      ps.flags |= kPrevValid;
    } else {
      PT_PROF(PF_DIFFUSE);
      // optix_device.cu:590-658: diffuse lobe, optionally guided
      f3 newDir = mk3(0.f);
      float cosTheta = 0.f, p_guided = 0.f, p_bsdf = 0.f;
      f3 lobe_rgb = mk3(1.0f);
      const GuideLobeDev* lobes = nullptr;
      uint32_t lobe_num = 0;
      float lobe_sum = 0.f;
      bool have_guiding = false;
      if (GUIDED && sp.guiding) {
        const GuideRegionDev R = guide_cell;
        if (R.lobe_num > 0) {
          lobes = gv.lobes + R.lobe_ofs;
          lobe_num = R.lobe_num;
          lobe_sum = R.sum;
          have_guiding = true;
        }
      }
      bool used_guiding = false;
      if (have_guiding) {
        const bool batch = lobe_num <= kGuideBatch;
        if (!PT_GUIDE_EARLY_BATCH && batch) guiding_load_batch(lobes, lobe_num, gb);
        const float u_lobe = next_rand(ps, rc);
        const int lobe_idx = batch ? guiding_choose_lobe_batch(gb, lobe_num, lobe_sum, u_lobe)
                                   : guiding_choose_lobe(lobes, lobe_num, lobe_sum, u_lobe);
        if (lobe_idx >= 0) {
          const float ux = next_rand(ps, rc);
          const float uy = next_rand(ps, rc);
          const GuideLobeDev lb = lobes[lobe_idx];
          GuideAxes ga;
          if (batch) guiding_load_axes(lobes, lobe_num, ga);  // in flight together with the chosen lobe's record
          const f3 gdir = vmf_sample(lb, ux, uy);
          const float cosG = fmaxf(dot(gdir, Ng), 0.0f);
          const float p_g = batch ? guiding_mixture_pdf_batch(ga, gb, lobe_num, lobe_sum, gdir)
                                  : guiding_mixture_pdf(lobes, lobe_num, lobe_sum, gdir);
          const float p_b = cosG * kInvPi;
          const float ch2 = next_rand(ps, rc) * (p_g + p_b);
          if (ch2 < p_g) {
            used_guiding = true;
            newDir = gdir; cosTheta = cosG; p_guided = p_g; p_bsdf = p_b;
            lobe_rgb = ld3(lb.rgb);
          }
        }
      }
      if (!used_guiding) {
        f3 local, tangent, bitangent;
        if (shared_disk) {
          cosTheta = fsqrt(1.0f - disk_r2);
          local = mk3(disk_x, disk_y, cosTheta);
          tangent = onb_t;
          bitangent = onb_b;
        } else {
          const float r1 = next_rand(ps, rc);
          const float r2 = next_rand(ps, rc);
          cosTheta = fsqrt(1.0f - r2);
          const float sinTheta = fsqrt(r2);
          local = mk3(cos2pi(r1) * sinTheta, sin2pi(r1) * sinTheta, cosTheta);
          make_onb(Ng, tangent, bitangent);
        }
        newDir = unit_by_construction(tangent * local.x + bitangent * local.y + Ng * local.z);
        p_bsdf = cosTheta * kInvPi;
        // (the reference also evaluates the mixture pdf of this direction, optix_device.cu:647, and never uses it)
      }
      ps.o = P + Ng * 1e-3f;
      ps.d = newDir;
      const float final_pdf = used_guiding ? p_guided : p_bsdf;
      f3 T = ps.T * kd;
      if (used_guiding) T = T * lobe_rgb;
      ps.T = T * (cosTheta * rcp(kPi * final_pdf * diff_prob));
      ps.prev_pdf = final_pdf * diff_prob;
      ps.flags |= kPrevValid;
    }
    // NEE contribution is weighted below, once the post-bounce throughput is known
    if (sr.valid) sr.contrib = Lo;
  }

  // the new ray leaves this primitive's plane if it points to the side its origin was pushed to
  if (dot(ps.d, ps.o - P) > 0.0f) ps.origin = sf.origin_hint;

  if (!depth_and_roulette(ps, rc, sp.max_depth)) ps.flags &= ~kAlive;

  // optix_device.cu:988-989: contribution = radiance * throughput (post-bounce)
  ps.sum += radiance * ps.T;
  if (sr.valid) sr.contrib = sr.contrib * ps.T;
}

// One bounce.  hit.prim < 0 is the radiance miss (optix_device.cu:370-375).
// GUIDED compiles the guided diffuse branch in (used when a field snapshot is active);
// TRAIN fills `tv` for the caller to append to the training ring.
template <bool GUIDED, bool TRAIN>
PT_HD void shade_hit(const SceneView& sc, const GuideView& gv, const ShadeParams& sp, const RandCtx& rc,
                     PathState& ps, const Hit& hit, ShadowRay& sr, TrainVertex& tv) {
  sr.valid = false;
  sr.origin = 0u;
  ps.origin = 0u;
  if (TRAIN) tv.valid = false;
  if (hit.prim < 0) {
    ps.flags &= ~kAlive;  // radiance = 0, done = 1: nothing to add
    return;
  }
  shade_surface<GUIDED, TRAIN>(sc, gv, sp, rc, ps, load_surface(sc, hit), sr, tv);
}
// ... of a scene made of instances: `inst` is the instance the hit primitive belongs to
template <bool GUIDED, bool TRAIN>
PT_HD void shade_hit_instanced(const SceneView& sc, const GuideView& gv, const ShadeParams& sp, const RandCtx& rc,
                               PathState& ps, const Hit& hit, uint32_t inst, ShadowRay& sr, TrainVertex& tv) {
  sr.valid = false;
  sr.origin = 0u;
  ps.origin = 0u;
  if (TRAIN) tv.valid = false;
  if (hit.prim < 0) {
    ps.flags &= ~kAlive;
    return;
  }
  shade_surface<GUIDED, TRAIN>(sc, gv, sp, rc, ps, load_surface_instanced(sc, hit, inst, ps.o, ps.d), sr, tv);
}

// optix_device.cu:955-969: start sample `s` of pixel (x,y).  The caller supplies the PCG
// state the sample starts from.
PT_HD void generate_path(const CameraView& cam, const RandCtx& rc, int s, int width, int height, uint64_t seed,
                         PathState& ps) {
  const int x = rc.px, y = rc.py;
  float u, v;
  blue_noise_shifted(rc.rx, rc.ry, s, u, v);
  const float jx = sample_tri(u), jy = sample_tri(v);  // optix_device.cu:192-196
  ps.o = ld3(cam.eye);
  ps.d = camera_dir(cam, x, y, jx, jy, width, height);
  ps.T = mk3(1.0f);
  ps.sum = mk3(0.0f);
  ps.seed = seed;
  ps.prev_pdf = 0.0f;
  ps.flags = kAlive | kFoldPending;
  ps.origin = 0u;
}

}  // namespace pt

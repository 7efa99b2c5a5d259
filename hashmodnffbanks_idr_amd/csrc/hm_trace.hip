// hm_trace.hip - sync-free ray/surface intersection search for gfx950: IDR's bidirectional
// sphere tracing, sign-change sampler + secant refinement and the closest-approach search,
// restated as PER-RAY STATE MACHINES driven by small update kernels around the fused SDF kernel.
//
// Replaces RayTracing.forward / sphere_tracing / ray_sampler / secant / minimal_sdf_points
// (reference: model/ray_tracing.py:26-298).  Every update in the reference is gated by a per-ray
// mask, so each ray can run its own state machine; the reference's
// host-side control flow (>= 9 device->host syncs, boolean-mask gathers with data-dependent
// shapes) disappears: each round is  [advance rays, append the points they need to a compact
// buffer with an atomic cursor]  ->  [fused SDF kernel over min(capacity, *cursor) points, the
// count read on the device].  One C call enqueues the whole search; nothing synchronises.
//
// Arithmetic is kept bit-compatible with the reference's elementwise fp32 expressions
// (p = cam + t*dir as mul-then-add, t +/- sdf, the secant formula's evaluation order); the file
// is compiled with -ffp-contract=off.
#include "hm_common.h"

#include <stdlib.h>

// hm_sdf.hip (not exported): rounds [first, rounds) of the sphere-tracing march as one persistent launch
extern "C" int hm_trace_march_tail(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table,
                                   const float *B_fourier, int frac_mode, int body16, const void *trace_args, int first,
                                   int rounds, void *stream);

extern "C" int hm_trace_secant_persistent(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table,
                                          const float *B_fourier, int frac_mode, int tile_points, const void *trace_args,
                                          int n_iters, void *stream);

// hm_sdf.hip (not exported): the closest-approach scan and the secant refinement as ONE launch (workgroups without secant
// rays start on the scan at once, 64-point tiles handed out by an atomic cursor)
extern "C" int hm_trace_scan_secant(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table,
                                    const float *B_fourier, int frac_mode, const void *trace_args, int n_iters,
                                    int64_t scan_off, int64_t scan_capacity, int c_prev1, int grid1, int c_prev2,
                                    int grid2, void *stream);

// hm_sdf.hip (not exported): hm_sdf_fwd whose 64-point launch completes its last round with tiles of a second point set
extern "C" int hm_sdf_fwd_fill(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *x, int64_t n,
                               const float *table, const float *B_fourier, float *out, int frac_mode, const int32_t *n_dev,
                               const float *x_fill, float *out_fill, const int32_t *n_fill_dev, const int32_t *prev_dev,
                               int prev_grid, int *grid64_out, void *stream);

namespace {

constexpr int kTB = 256;

#include "hm_trace_dev.h"

// round 0: both sphere intersections of every ray that hits the bounding sphere (ray_tracing.py:101-128)
__global__ __launch_bounds__(kTB) void trace_init_kernel(TraceArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= a.n) return;
    trace_init_ray(a, i, a.w.cnt + C_ROUND0, 0);
}

// one round of the per-ray sphere-tracing state machine (ray_tracing.py:130-186): the points it appends are evaluated
// by the next round's SDF launch
__global__ __launch_bounds__(kTB) void trace_advance_kernel(TraceArgs a, int round) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= a.n) return;
    trace_advance_ray(a, i, a.w.cnt + C_ROUND0 + round, 0);
}

// after the march: network mask, provisional outputs, hand unconverged rays to the sampler
__global__ __launch_bounds__(kTB) void trace_finalize_kernel(TraceArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= a.n) return;
    const TraceWs &w = a.w;
    if (w.stage[i] != ST_DONE) atomicAdd(w.cnt + C_UNFINISHED, 1);
    const float t_s = w.t_s[i], t_e = w.t_e[i];
    a.out_mask[i] = t_s < t_e;
    a.out_t[i] = t_s;
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    // the reference recomputes cam + t*dir for ALL rays once its loop has run at least one iteration
    if (a.hit[i] || w.cnt[C_ANY_LIVE]) along(a, i, t_s, px, py, pz);
    a.out_pts[i * 3] = px; a.out_pts[i * 3 + 1] = py; a.out_pts[i * 3 + 2] = pz;
    const bool samp = w.live_s[i] != 0;
    w.is_samp[i] = samp;
    if (samp) w.list_samp[atomicAdd(w.cnt + C_NSAMP, 1)] = (int32_t)i;
}

// n_steps samples along every listed ray: t = lo + f*(hi - lo)  (ray_tracing.py:198-206, 277-286)
// c_base >= 0: the points are appended behind the cnt[c_base] points already in the buffer (one SDF launch then
// evaluates both sets); cnt[C_BIG_PTS] receives the total.
//
// Lazy sampler (a.head > 0): the reference reads, of a sampler ray's n_steps values, only those up to its FIRST negative
// sample (argmin of sign * (n, n-1, ..), ray_tracing.py:212-218, plus the sample before it - or the last one when the
// first sample is negative - for the secant bracket, :238-243) unless the ray falls back to the minimal sample
// (:221-226), which needs all of them.  per_ray < n_steps selects the first pass: samples 0..head-1 and n_steps-1.
__global__ __launch_bounds__(kTB) void ray_samples_kernel(TraceArgs a, const int32_t *list, int c_n, int c_npts,
                                                          const float *lo_arr, const float *hi_arr,
                                                          const float *fr, int c_base, int per_ray) {
    const int64_t gid = (int64_t)blockIdx.x * kTB + threadIdx.x;
    const TraceWs &w = a.w;
    const int32_t n_list = w.cnt[c_n];
    const int64_t base = c_base == -2 ? a.sel_off : (c_base >= 0 ? w.cnt[c_base] : 0);   // (-2: the region of its own)
    if (gid == 0) {
        w.cnt[c_npts] = n_list * per_ray;
        if (c_base != -2) w.cnt[C_BIG_PTS] = (int32_t)base + n_list * per_ray;
    }
    const int64_t m = gid / per_ray;
    if (m >= n_list) return;
    int s = (int)(gid - m * per_ray);
    if (per_ray < a.n_steps && s == per_ray - 1) s = a.n_steps - 1;
    const int64_t i = list[m];
    const float lo = lo_arr[i], hi = hi_arr[i];
    const float t = __fadd_rn(lo, __fmul_rn(fr[s], __fsub_rn(hi, lo)));
    float px, py, pz;
    along(a, i, t, px, py, pz);
    const int64_t o = base + gid;
    w.pts[o * 3] = px; w.pts[o * 3 + 1] = py; w.pts[o * 3 + 2] = pz;
}

// lazy sampler, between the passes: a ray is resolved by its head samples iff one of them is negative and the ray is
// inside the object mask; every other ray joins the second pass (samples head..n_steps-2)
__global__ __launch_bounds__(kTB) void sampler_head_kernel(TraceArgs a) {
    const int64_t m = (int64_t)blockIdx.x * kTB + threadIdx.x;
    const TraceWs &w = a.w;
    if (m >= w.cnt[C_NSAMP]) return;
    const int64_t i = w.list_samp[m];
    const float *v = w.vals + m * (a.head + 1);
    bool neg = false;
    for (int s = 0; s < a.head; ++s) neg = neg || (v[s] < 0.0f);
    const bool resolved = neg && a.obj[i] != 0;
    w.tail_slot[m] = resolved ? -1 : atomicAdd(w.cnt + C_NSAMP2, 1);
}

__global__ __launch_bounds__(kTB) void sampler_tail_points_kernel(TraceArgs a) {
    const int64_t gid = (int64_t)blockIdx.x * kTB + threadIdx.x;
    const TraceWs &w = a.w;
    const int per = a.n_steps - a.head - 1;
    if (gid == 0) {
        w.cnt[C_TAIL_PTS] = w.cnt[C_NSAMP2] * per;
        w.cnt[C_NSAMP_PTS] = w.cnt[C_HEAD_PTS] + w.cnt[C_NSAMP2] * per;     // statistics: sampler evaluations
    }
    const int64_t m = gid / per;
    if (m >= w.cnt[C_NSAMP]) return;
    const int32_t q = w.tail_slot[m];
    if (q < 0) return;
    const int j = (int)(gid - m * per);
    const int64_t i = w.list_samp[m];
    const float lo = w.t_s[i], hi = w.t_e[i];
    const float t = __fadd_rn(lo, __fmul_rn(a.fracs[a.head + j], __fsub_rn(hi, lo)));
    float px, py, pz;
    along(a, i, t, px, py, pz);
    const int64_t o = a.tail_off + (int64_t)q * per + j;
    w.pts[o * 3] = px; w.pts[o * 3 + 1] = py; w.pts[o * 3 + 2] = pz;
}

// first sign change / minimal sample per sampler ray, secant bracket set-up (ray_tracing.py:212-247)
__global__ __launch_bounds__(kTB) void sampler_reduce_kernel(TraceArgs a) {
    const int64_t m = (int64_t)blockIdx.x * kTB + threadIdx.x;
    const TraceWs &w = a.w;
    if (m >= w.cnt[C_NSAMP]) return;
    const int64_t i = w.list_samp[m];
    const int n = a.n_steps;
    // value of sample s: one [n_steps] row (single pass) or head row + second-pass row (lazy sampler)
    const bool lazy = a.head > 0;
    const int32_t tq = lazy ? w.tail_slot[m] : 0;    // second-pass slot, -1 = resolved by the head samples
    const float *v_head = lazy ? w.vals + m * (a.head + 1) : w.vals + m * n;
    const float *v_tail = lazy ? w.vals + a.tail_off + (int64_t)(tq < 0 ? 0 : tq) * (n - a.head - 1) - a.head : v_head;
    const int head = a.head;
    auto v = [&](int s) -> float {
        if (!lazy) return v_head[s];
        if (s < head) return v_head[s];
        return s == n - 1 ? v_head[head] : v_tail[s];
    };
    // a ray resolved by its head samples has its first negative sample there: later samples cannot change `first`
    // (their sign * rank is larger) and nothing else is read from them
    const int n_scan = (lazy && tq < 0) ? head : n;
    int first = 0, amin = 0, bad = 0;
    float best_tmp = 0.0f, best_v = 0.0f;
    auto visit = [&](int s, float x) {
        bad += !isfinite(x);
        const float sg = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
        const float tmp = sg * (float)(n - s);  // sign(sdf) * arange(n, 0, -1)
        if (s == 0 || tmp < best_tmp) { best_tmp = tmp; first = s; }
        if (s == 0 || x < best_v) { best_v = x; amin = s; }
    };
    if (!lazy && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(v_head) & 15u) == 0) {
        // single-pass rows are n contiguous floats: 16-byte loads, several in flight - the scalar loop below waits for
        // one 4-byte load per sample (a serial L2 round trip each: 42 us for 1281 rays of 100 samples)
        const float4 *row = reinterpret_cast<const float4 *>(v_head);
#pragma unroll 5
        for (int g4 = 0; g4 < n / 4; ++g4) {
            const float4 x4 = row[g4];
            visit(4 * g4, x4.x); visit(4 * g4 + 1, x4.y); visit(4 * g4 + 2, x4.z); visit(4 * g4 + 3, x4.w);
        }
    } else {
        for (int s = 0; s < n_scan; ++s) visit(s, v(s));
    }
    const float lo = w.t_s[i], hi = w.t_e[i];
    auto t_at = [&](int s) { return __fadd_rn(lo, __fmul_rn(a.fracs[s], __fsub_rn(hi, lo))); };
    const float v_first = v(first);
    const bool net_hit = v_first < 0.0f;
    const bool true_obj = a.obj[i] != 0;
    const int pick = (true_obj && net_hit) ? first : amin;
    float t_out = t_at(pick);
    float px, py, pz;
    along(a, i, t_out, px, py, pz);
    a.out_mask[i] = net_hit;
    const bool sec = a.training ? (net_hit && true_obj) : net_hit;
    if (sec) {
        const int lo_idx = first > 0 ? first - 1 : n - 1;  // index -1 wraps to the last sample, as in the reference
        const float z_hi = t_at(first), v_hi = v_first, z_lo = t_at(lo_idx), v_lo = v(lo_idx);
        if (lazy && tq < 0 && lo_idx == n - 1 && !isfinite(v_lo)) ++bad;   // (outside the scanned range)
        const float z = secant_z(v_lo, v_hi, z_lo, z_hi);
        const int32_t q = atomicAdd(w.cnt + C_NSEC, 1);
        w.list_sec[q] = (int32_t)i;
        w.z_lo[q] = z_lo; w.z_hi[q] = z_hi; w.v_lo[q] = v_lo; w.v_hi[q] = v_hi; w.z[q] = z;
        if (a.n_secant == 0) {  // no refinement rounds: the initial secant estimate is the answer
            t_out = z;
            along(a, i, z, px, py, pz);
        }
    }
    if (bad) atomicAdd(w.cnt + C_NONFINITE, bad);
    a.out_t[i] = t_out;
    w.t_s[i] = t_out;
    a.out_pts[i * 3] = px; a.out_pts[i * 3 + 1] = py; a.out_pts[i * 3 + 2] = pz;
}

// p_mid = cam + z*dir for every secant ray (the points of the next SDF launch)
__global__ __launch_bounds__(kTB) void secant_points_kernel(TraceArgs a) {
    const int64_t q = (int64_t)blockIdx.x * kTB + threadIdx.x;
    const TraceWs &w = a.w;
    if (q >= w.cnt[C_NSEC]) return;
    float px, py, pz;
    along(a, w.list_sec[q], w.z[q], px, py, pz);
    w.pts[q * 3] = px; w.pts[q * 3 + 1] = py; w.pts[q * 3 + 2] = pz;
}

// one secant iteration (ray_tracing.py:255-266), stand-alone form of secant_advance_ray
__global__ __launch_bounds__(kTB) void secant_advance_kernel(TraceArgs a, int last) {
    const int64_t q = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (q >= a.w.cnt[C_NSEC]) return;
    secant_advance_ray(a, q, last);
}

// training tail, part 1 (ray_tracing.py:71-88): rays that feed the mask loss
__global__ __launch_bounds__(kTB) void tail_prepare_kernel(TraceArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= a.n) return;
    const TraceWs &w = a.w;
    const bool net = a.out_mask[i] != 0, obj = a.obj[i] != 0, samp = w.is_samp[i] != 0, hit = a.hit[i] != 0;
    const bool in_mask = !net && obj && !samp;
    const bool out_mask = !obj && !samp;
    if (!(in_mask || out_mask)) return;
    if (!hit) {
        // the ray never enters the sphere: closest point of the ray to the origin, t = -<dir, cam>
        const float *c = a.cam + (i / a.rays_per_image) * 3;
        const float *d = a.dirs + i * 3;
        const float dot = __fmaf_rn(d[2], c[2], __fmaf_rn(d[1], c[1], __fmul_rn(d[0], c[0])));
        const float t = -dot;
        float px, py, pz;
        along(a, i, t, px, py, pz);
        a.out_t[i] = t;
        w.t_s[i] = t;
        a.out_pts[i * 3] = px; a.out_pts[i * 3 + 1] = py; a.out_pts[i * 3 + 2] = pz;
        return;
    }
    if (net && out_mask) w.t_min[i] = w.t_s[i];
    w.list_sel[atomicAdd(w.cnt + C_NSEL, 1)] = (int32_t)i;
}

// training tail, part 2 (ray_tracing.py:288-296): argmin of the SDF over the shared random fractions
__global__ __launch_bounds__(kTB) void closest_reduce_kernel(TraceArgs a) {
    const int64_t m = (int64_t)blockIdx.x * kTB + threadIdx.x;
    const TraceWs &w = a.w;
    if (m >= w.cnt[C_NSEL]) return;
    const int64_t i = w.list_sel[m];
    const int n = a.n_steps;
    const float *v = w.vals + (a.sel_off >= 0 ? a.sel_off : (int64_t)w.cnt[a.head > 0 ? C_HEAD_PTS : C_NSAMP_PTS]) +
                     m * n;   // a region of their own, or behind the sampler's first pass
    int amin = 0, bad = !isfinite(v[0]);
    float best = v[0];
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15u) == 0) {   // (16-byte loads: see sampler_reduce_kernel)
        const float4 *row = reinterpret_cast<const float4 *>(v);
        auto visit = [&](int s, float x) {
            if (s == 0) return;
            bad += !isfinite(x);
            if (x < best) { best = x; amin = s; }
        };
#pragma unroll 5
        for (int g4 = 0; g4 < n / 4; ++g4) {
            const float4 x4 = row[g4];
            visit(4 * g4, x4.x); visit(4 * g4 + 1, x4.y); visit(4 * g4 + 2, x4.z); visit(4 * g4 + 3, x4.w);
        }
    } else {
        for (int s = 1; s < n; ++s) {
            bad += !isfinite(v[s]);
            if (v[s] < best) { best = v[s]; amin = s; }
        }
    }
    if (bad) atomicAdd(w.cnt + C_NONFINITE, bad);
    const float lo = w.t_min[i], hi = w.t_max[i];
    const float t = __fadd_rn(__fmul_rn(a.steps_u[amin], __fsub_rn(hi, lo)), lo);
    float px, py, pz;
    along(a, i, t, px, py, pz);
    a.out_t[i] = t;
    a.out_pts[i * 3] = px; a.out_pts[i * 3 + 1] = py; a.out_pts[i * 3 + 2] = pz;
}

__global__ void count_evals_kernel(int32_t *cnt, int rounds, int n_secant) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t tot = 0;
    for (int r = 0; r < rounds && r < 64; ++r) tot += cnt[C_ROUND0 + r];
    tot += cnt[C_NSAMP_PTS] + (int64_t)cnt[C_NSEC] * n_secant + cnt[C_NSEL_PTS];
    cnt[C_EVALS] = (int32_t)tot;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
    size_t off_f, off_i, off_b, off_pts, off_vals, off_cnt, off_emb, total;
    int64_t cap;   // points of one SDF launch; pts / vals hold two such regions (the lazy sampler's second pass)
};

Layout make_layout(int64_t n, int n_steps, int emb_width = 0) {
    Layout L;
    L.cap = n * (n_steps > 2 ? n_steps : 2);
    size_t o = 0;
    L.off_f = o; o = align_up(o + sizeof(float) * 13 * (size_t)n, 256);
    L.off_i = o; o = align_up(o + sizeof(int32_t) * 6 * (size_t)n, 256);
    L.off_b = o; o = align_up(o + 6 * (size_t)n, 256);
    L.off_pts = o; o = align_up(o + sizeof(float) * 3 * 3 * (size_t)L.cap, 256);    // three regions of `cap` points
    L.off_vals = o; o = align_up(o + sizeof(float) * 3 * (size_t)L.cap, 256);
    L.off_cnt = o; o = align_up(o + sizeof(int32_t) * C_COUNT, 256);
    L.off_emb = o; o = align_up(o + sizeof(float) * (size_t)emb_width * (size_t)L.cap, 256);   // (filter-bank embedders)
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

int64_t hm_trace_workspace_bytes(int64_t n_rays, const hm_trace_cfg *cfg) {
    if (n_rays < 0 || !cfg || cfg->n_steps < 1) return hm_fail(HM_ERR_INVALID, "hm_trace_workspace_bytes: bad argument");
    return (int64_t)make_layout(n_rays, cfg->n_steps).total;
}

int64_t hm_trace_workspace_bytes_nffb(int64_t n_rays, const hm_trace_cfg *cfg, int n_levels) {
    if (n_rays < 0 || !cfg || cfg->n_steps < 1 || n_levels < 1 || n_levels > HM_MAX_LEVELS)
        return hm_fail(HM_ERR_INVALID, "hm_trace_workspace_bytes_nffb: bad argument");
    return (int64_t)make_layout(n_rays, cfg->n_steps, 3 + 8 + 8 * n_levels).total;
}

static int trace_forward_impl(const hm_grid_desc *desc, const hm_nffb_desc *nffb, const hm_mlp_desc *mlp,
                              const float *table, const float *B_fourier, int frac_mode, int tile_points,
                              const hm_trace_cfg *cfg, const float *cam_loc, const float *ray_dirs,
                              const uint8_t *object_mask, const float *t_sphere, const uint8_t *hit_mask, int64_t n_rays,
                              int64_t rays_per_image, const float *sampler_fracs, const float *steps_u,
                              float *out_points, uint8_t *out_net_mask, float *out_dists, void *workspace,
                              int64_t workspace_bytes, int32_t *stats_out, void *stream);

int hm_trace_forward(const hm_grid_desc *desc, const hm_mlp_desc *mlp, const float *table, const float *B_fourier,
                     int frac_mode, int tile_points, const hm_trace_cfg *cfg, const float *cam_loc,
                     const float *ray_dirs, const uint8_t *object_mask, const float *t_sphere,
                     const uint8_t *hit_mask, int64_t n_rays, int64_t rays_per_image, const float *sampler_fracs,
                     const float *steps_u, float *out_points, uint8_t *out_net_mask, float *out_dists,
                     void *workspace, int64_t workspace_bytes, int32_t *stats_out, void *stream) {
    return trace_forward_impl(desc, nullptr, mlp, table, B_fourier, frac_mode, tile_points, cfg, cam_loc, ray_dirs,
                              object_mask, t_sphere, hit_mask, n_rays, rays_per_image, sampler_fracs, steps_u, out_points,
                              out_net_mask, out_dists, workspace, workspace_bytes, stats_out, stream);
}

int hm_trace_forward_nffb(const hm_grid_desc *desc, const hm_nffb_desc *nffb, const hm_mlp_desc *mlp, const float *table,
                          const float *B_fourier, int frac_mode, int tile_points, const hm_trace_cfg *cfg,
                          const float *cam_loc, const float *ray_dirs, const uint8_t *object_mask, const float *t_sphere,
                          const uint8_t *hit_mask, int64_t n_rays, int64_t rays_per_image, const float *sampler_fracs,
                          const float *steps_u, float *out_points, uint8_t *out_net_mask, float *out_dists,
                          void *workspace, int64_t workspace_bytes, int32_t *stats_out, void *stream) {
    HM_CHECK_ARG(nffb != nullptr, "hm_trace_forward_nffb: NULL embedder descriptor");
    return trace_forward_impl(desc, nffb, mlp, table, B_fourier, frac_mode, tile_points, cfg, cam_loc, ray_dirs,
                              object_mask, t_sphere, hit_mask, n_rays, rays_per_image, sampler_fracs, steps_u, out_points,
                              out_net_mask, out_dists, workspace, workspace_bytes, stats_out, stream);
}

static int trace_forward_impl(const hm_grid_desc *desc, const hm_nffb_desc *nffb, const hm_mlp_desc *mlp,
                              const float *table, const float *B_fourier, int frac_mode, int tile_points,
                              const hm_trace_cfg *cfg, const float *cam_loc, const float *ray_dirs,
                              const uint8_t *object_mask, const float *t_sphere, const uint8_t *hit_mask, int64_t n_rays,
                              int64_t rays_per_image, const float *sampler_fracs, const float *steps_u,
                              float *out_points, uint8_t *out_net_mask, float *out_dists, void *workspace,
                              int64_t workspace_bytes, int32_t *stats_out, void *stream) {
    HM_CHECK_ARG(desc && mlp && cfg, "hm_trace_forward: NULL descriptor");
    HM_CHECK_ARG(n_rays >= 0 && rays_per_image >= 1, "hm_trace_forward: bad ray counts");
    HM_CHECK_ARG(cfg->n_steps >= 2 && cfg->n_steps <= 4096, "hm_trace_forward: n_steps out of range");
    HM_CHECK_ARG(cfg->sphere_tracing_iters >= 0 && cfg->sphere_tracing_iters <= 15 && cfg->line_step_iters >= 0 &&
                     cfg->line_step_iters <= 3 &&
                     1 + cfg->sphere_tracing_iters * (1 + cfg->line_step_iters) <= 64,
                 "hm_trace_forward: too many march rounds (sphere_tracing_iters <= 15, line_step_iters <= 3)");
    HM_CHECK_ARG(cfg->n_secant_steps >= 0 && cfg->n_secant_steps <= 64, "hm_trace_forward: n_secant_steps out of range");
    if (n_rays == 0) return HM_OK;
    HM_CHECK_ARG(n_rays * (int64_t)cfg->n_steps < (1ll << 31), "hm_trace_forward: too many rays for one call");
    HM_CHECK_ARG(table && B_fourier && cam_loc && ray_dirs && object_mask && t_sphere && hit_mask && sampler_fracs &&
                     out_points && out_net_mask && out_dists && workspace,
                 "hm_trace_forward: NULL pointer");
    HM_CHECK_ARG(!cfg->training || steps_u, "hm_trace_forward: training mode needs steps_u");
    const int emb_width = nffb ? 3 + 8 + 8 * nffb->n_levels : 0;
    const Layout L = make_layout(n_rays, cfg->n_steps, emb_width);
    HM_CHECK_ARG(workspace_bytes >= (int64_t)L.total, "hm_trace_forward: workspace too small");
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(workspace);
    const size_t n = (size_t)n_rays;
    TraceArgs a;
    float *f = reinterpret_cast<float *>(base + L.off_f);
    a.w.t_s = f; a.w.t_e = f + n; a.w.t_min = f + 2 * n; a.w.t_max = f + 3 * n; a.w.cur_s = f + 4 * n;
    a.w.cur_e = f + 5 * n; a.w.nxt_s = f + 6 * n; a.w.nxt_e = f + 7 * n; a.w.z_lo = f + 8 * n; a.w.z_hi = f + 9 * n;
    a.w.v_lo = f + 10 * n; a.w.v_hi = f + 11 * n; a.w.z = f + 12 * n;
    int32_t *ip = reinterpret_cast<int32_t *>(base + L.off_i);
    a.w.slot_s = ip; a.w.slot_e = ip + n; a.w.list_samp = ip + 2 * n; a.w.list_sec = ip + 3 * n; a.w.list_sel = ip + 4 * n;
    a.w.tail_slot = ip + 5 * n;
    uint8_t *bp = reinterpret_cast<uint8_t *>(base + L.off_b);
    a.w.live_s = bp; a.w.live_e = bp + n; a.w.stage = bp + 2 * n; a.w.it = bp + 3 * n; a.w.k = bp + 4 * n;
    a.w.is_samp = bp + 5 * n;
    a.w.pts = reinterpret_cast<float *>(base + L.off_pts);
    a.w.vals = reinterpret_cast<float *>(base + L.off_vals);
    a.w.cnt = reinterpret_cast<int32_t *>(base + L.off_cnt);
    a.w.cap = L.cap;
    a.cam = cam_loc; a.dirs = ray_dirs; a.obj = object_mask; a.t_sphere = t_sphere; a.hit = hit_mask;
    a.fracs = sampler_fracs; a.steps_u = steps_u;
    a.out_pts = out_points; a.out_mask = out_net_mask; a.out_t = out_dists;
    a.n = n_rays; a.rays_per_image = rays_per_image;
    a.thr = cfg->sdf_threshold;
    for (int k = 0; k < 4; ++k) a.back[k] = (float)((1.0 - cfg->line_search_step) / (double)(1 << k));
    a.ls_iters = cfg->line_step_iters; a.max_it = cfg->sphere_tracing_iters; a.n_steps = cfg->n_steps;
    a.n_secant = cfg->n_secant_steps; a.training = cfg->training;
    // lazy sampler: a first pass over samples 0..head-1 and n_steps-1 only pays when it leaves something out
    a.head = (cfg->sampler_head >= 1 && cfg->sampler_head + 1 < cfg->n_steps) ? cfg->sampler_head : 0;
    a.tail_off = L.cap;
    // Closest-approach scan and secant refinement as one launch (hm_sdf.hip: sdf_scan_secant_kernel): training, hash-grid
    // network, exact-fp32 coarse scans, tile size left to the library.  HM_TRACE_OVERLAP=0: the scan joins the sampler's
    // launch and the secant runs behind it on a few dozen workgroups (A/B).
    const char *e_ov = getenv("HM_TRACE_OVERLAP");      // (read per call: the tests compare both forms in one process)
    const bool overlap_ok = !(e_ov && atoi(e_ov) == 0);
    static const bool persistent_ok = [] { const char *e = getenv("HM_TRACE_PERSISTENT"); return !(e && atoi(e) == 0); }();
    const bool overlap = overlap_ok && persistent_ok && cfg->training && !nffb && !cfg->coarse_bf16 && tile_points == 0 &&
                         cfg->n_secant_steps > 0 && n_rays <= 8192;
    a.sel_off = overlap ? 2 * L.cap : -1;

    hm_zero_u32_async(a.w.cnt, C_COUNT, st);
    const unsigned g_rays = (unsigned)((n_rays + kTB - 1) / kTB);
    const unsigned g_samp = (unsigned)((n_rays * cfg->n_steps + kTB - 1) / kTB);

    float *emb_ws = reinterpret_cast<float *>(base + L.off_emb);
    auto sdf = [&](int64_t capacity, const int32_t *n_dev) -> int {
        if (nffb) {   // filter-bank embedder: its own fused kernel, then the MLP on the embedding rows
            const int rc = hm_nffb_fwd(desc, nffb, a.w.pts, capacity, table, B_fourier, emb_ws, emb_width, frac_mode,
                                       n_dev, stream);
            if (rc != HM_OK) return rc;
            return hm_sdf_fwd_emb(mlp, emb_ws, emb_width, emb_width, capacity, a.w.vals, 1, 1, tile_points, n_dev, 0,
                                  stream);
        }
        return hm_sdf_fwd(desc, mlp, a.w.pts, capacity, table, B_fourier, a.w.vals, 1, 1, frac_mode, tile_points, n_dev,
                          0, stream);
    };
    // the coarse scans (sampler passes, closest approach): `off` = first point of the region in pts / vals
    auto coarse = [&](int64_t off, int64_t capacity, const int32_t *n_dev) -> int {
        const float *p = a.w.pts + off * 3;
        float *v = a.w.vals + off;
        int rc;
        if (!cfg->coarse_bf16) {
            if (nffb) {
                rc = hm_nffb_fwd(desc, nffb, p, capacity, table, B_fourier, emb_ws, emb_width, frac_mode, n_dev, stream);
                if (rc == HM_OK)
                    rc = hm_sdf_fwd_emb(mlp, emb_ws, emb_width, emb_width, capacity, v, 1, 1, tile_points, n_dev, 0, stream);
            } else {
                rc = hm_sdf_fwd(desc, mlp, p, capacity, table, B_fourier, v, 1, 1, frac_mode, tile_points, n_dev, 0, stream);
            }
        } else if (nffb) {   // coarse scans on the 16-bit matrix cores above 8192 live points, exact fp32 small tiles below
            rc = hm_nffb_fwd(desc, nffb, p, capacity, table, B_fourier, emb_ws, emb_width, frac_mode, n_dev, stream);
            if (rc == HM_OK)
                rc = hm_sdf_fwd_emb(mlp, emb_ws, emb_width, emb_width, capacity, v, 1, 1, -1, n_dev, 0, stream);
            if (rc == HM_OK)
                rc = cfg->coarse_bf16 == 2
                         ? hm_sdf_fwd_emb_split(mlp, emb_ws, emb_width, emb_width, capacity, v, 1, n_dev, 8193, stream)
                         : hm_sdf_fwd_emb_bf16(mlp, emb_ws, emb_width, emb_width, capacity, v, 1, n_dev, 8193, stream);
        } else {
            rc = hm_sdf_fwd(desc, mlp, p, capacity, table, B_fourier, v, 1, 1, frac_mode, -1, n_dev, 0, stream);
            if (rc == HM_OK)
                rc = cfg->coarse_bf16 == 2
                         ? hm_sdf_fwd_split(desc, mlp, p, capacity, table, B_fourier, v, 1, frac_mode, n_dev, 8193, stream)
                         : hm_sdf_fwd_bf16(desc, mlp, p, capacity, table, B_fourier, v, 1, frac_mode, n_dev, 8193, stream);
        }
        return rc;
    };

    // ---- 1. bidirectional sphere tracing: one state-machine round per SDF launch -------------------
    // 1 + sphere_tracing_iters rounds are needed when no ray runs a line search, `rounds` when one does in every
    // iteration.  Hash-grid networks with the tile size left to the library (or fixed at 16) run the surplus rounds as
    // ONE persistent launch in which every workgroup carries eight rays to the end (hm_sdf.hip: trace_march_tail_kernel;
    // it returns at once when no ray is left) instead of 30 (empty SDF launch, empty update launch) pairs.
    // HM_TRACE_PERSISTENT=0: every round a launch pair (A/B).
    const int rounds = 1 + cfg->sphere_tracing_iters * (1 + cfg->line_step_iters);
    const bool tail = persistent_ok && !nffb && (tile_points == 0 || tile_points == 16) &&
                      a.w.cap >= ((n_rays + 7) / 8) * 16 && cfg->line_step_iters > 0;
    int launched = tail ? 1 + cfg->sphere_tracing_iters : rounds;
    if (tail) {   // HM_TRACE_TAIL_FIRST=k (tests): hand over to the persistent kernel after k rounds already (read per call)
        const char *e = getenv("HM_TRACE_TAIL_FIRST");
        const int k = e ? atoi(e) : 0;
        if (k >= 1 && k < launched) launched = k;
    }
    hipLaunchKernelGGL(trace_init_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
    for (int r = 0; r < launched; ++r) {
        int rc = sdf(2 * n_rays, a.w.cnt + C_ROUND0 + r);
        if (rc != HM_OK) return rc;
        // round r+1 cursor (the last advance appends nothing that is evaluated; it only closes states)
        hipLaunchKernelGGL(trace_advance_kernel, dim3(g_rays), dim3(kTB), 0, st, a, r + 1 < 64 ? r + 1 : 63);
    }
    if (tail) {
        const int rc = hm_trace_march_tail(desc, mlp, table, B_fourier, frac_mode, tile_points == 16 ? 1 : 0, &a, launched,
                                           rounds, stream);
        if (rc != HM_OK) return rc;
    }
    hipLaunchKernelGGL(trace_finalize_kernel, dim3(g_rays), dim3(kTB), 0, st, a);

    // ---- 2. sampler + (training) closest approach: ONE SDF launch over both point sets ----------------------
    // The mask-loss rays (ray_tracing.py:71-92) are exactly the rays the sampler does NOT touch, so their
    // n_steps random-fraction samples do not depend on the sampler's outcome: they are appended behind the
    // sampler's points and evaluated by the same launch (one dependent launch and one partial last wave less).
    if (cfg->training) hipLaunchKernelGGL(tail_prepare_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
    const int per_ray = a.head > 0 ? a.head + 1 : cfg->n_steps;    // samples per ray in the sampler's first pass
    // (single pass: the first pass IS the sampler's evaluation count; lazy: the second pass adds to it later)
    const int c_first = a.head > 0 ? (int)C_HEAD_PTS : (int)C_NSAMP_PTS;
    hipLaunchKernelGGL(ray_samples_kernel, dim3((unsigned)((n_rays * per_ray + kTB - 1) / kTB)), dim3(kTB), 0, st, a,
                       a.w.list_samp, (int)C_NSAMP, c_first, a.w.t_s, a.w.t_e, sampler_fracs, -1, per_ray);
    if (cfg->training)
        hipLaunchKernelGGL(ray_samples_kernel, dim3(g_samp), dim3(kTB), 0, st, a, a.w.list_sel, (int)C_NSEL,
                           (int)C_NSEL_PTS, a.w.t_min, a.w.t_max, steps_u, overlap ? -2 : c_first, cfg->n_steps);
    int grid_p1 = 0, grid_p2 = 0;      // overlap: grids of the sampler launches that took filler tiles of the scan
    if (overlap) {
        // the sampler's points only - the closest-approach scan runs with the secant refinement below, except for the
        // tiles that complete the last round of this launch (hm_sdf.hip: fill_quota)
        int rc = hm_sdf_fwd_fill(desc, mlp, a.w.pts, n_rays * cfg->n_steps, table, B_fourier, a.w.vals, frac_mode,
                                 a.w.cnt + c_first, a.w.pts + a.sel_off * 3, a.w.vals + a.sel_off, a.w.cnt + C_NSEL_PTS,
                                 nullptr, 0, &grid_p1, stream);
        if (rc != HM_OK) return rc;
    } else {
        int rc = coarse(0, n_rays * cfg->n_steps, a.w.cnt + C_BIG_PTS);
        if (rc != HM_OK) return rc;
    }
    if (cfg->training && !overlap) hipLaunchKernelGGL(closest_reduce_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
    if (a.head > 0) {   // second pass: the remaining samples of the rays the head samples did not resolve
        hipLaunchKernelGGL(sampler_head_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
        hipLaunchKernelGGL(sampler_tail_points_kernel, dim3(g_samp), dim3(kTB), 0, st, a);
        int rc = overlap ? hm_sdf_fwd_fill(desc, mlp, a.w.pts + a.tail_off * 3, n_rays * (cfg->n_steps - a.head - 1), table,
                                           B_fourier, a.w.vals + a.tail_off, frac_mode, a.w.cnt + C_TAIL_PTS,
                                           a.w.pts + a.sel_off * 3, a.w.vals + a.sel_off, a.w.cnt + C_NSEL_PTS,
                                           a.w.cnt + c_first, grid_p1, &grid_p2, stream)
                         : coarse(a.tail_off, n_rays * (cfg->n_steps - a.head - 1), a.w.cnt + C_TAIL_PTS);
        if (rc != HM_OK) return rc;
    }
    hipLaunchKernelGGL(sampler_reduce_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
    if (cfg->n_secant_steps > 0) {
        hipLaunchKernelGGL(secant_points_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
        // hash-grid networks with small-tile SDF launches (tile size left to the library, or 4 / 8 / 16) and a batch
        // small enough for them: all iterations as ONE launch in which a tile of secant rays stays with its workgroup
        // (hm_sdf.hip: trace_secant_kernel; with the closest-approach scan in the same launch: sdf_scan_secant_kernel);
        // otherwise an (SDF launch, update launch) pair per iteration
        if (overlap) {
            const int rc = hm_trace_scan_secant(desc, mlp, table, B_fourier, frac_mode, &a, cfg->n_secant_steps, a.sel_off,
                                                n_rays * cfg->n_steps, c_first, grid_p1, (int)C_TAIL_PTS, grid_p2, stream);
            if (rc != HM_OK) return rc;
            hipLaunchKernelGGL(closest_reduce_kernel, dim3(g_rays), dim3(kTB), 0, st, a);
        } else if (persistent_ok && !nffb && (tile_points == 0 || tile_points == 4 || tile_points == 8 || tile_points == 16) &&
            (tile_points != 0 || n_rays <= 8192)) {
            const int rc = hm_trace_secant_persistent(desc, mlp, table, B_fourier, frac_mode, tile_points, &a,
                                                      cfg->n_secant_steps, stream);
            if (rc != HM_OK) return rc;
        } else {
            for (int s = 0; s < cfg->n_secant_steps; ++s) {
                int rc = sdf(n_rays, a.w.cnt + C_NSEC);
                if (rc != HM_OK) return rc;
                hipLaunchKernelGGL(secant_advance_kernel, dim3(g_rays), dim3(kTB), 0, st, a,
                                   s == cfg->n_secant_steps - 1 ? 1 : 0);
            }
        }
    }
    hipLaunchKernelGGL(count_evals_kernel, dim3(1), dim3(64), 0, st, a.w.cnt, rounds + 1, cfg->n_secant_steps);
    if (stats_out) {
        hipLaunchKernelGGL(hm_copy_u32_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<uint32_t *>(stats_out),
                           reinterpret_cast<const uint32_t *>(a.w.cnt + C_NSAMP), 16);
    }
    HM_CHECK_LAUNCH("hm_trace_forward");
    return HM_OK;
}

}  // extern "C"

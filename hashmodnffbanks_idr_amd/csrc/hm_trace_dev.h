// hm_trace_dev.h - the ray search's per-ray state (TraceArgs) and the sphere-tracing state machine (start of a ray, one
// march round) as DEVICE FUNCTIONS, shared by hm_trace.hip (launch-per-round search: stand-alone update kernels around
// the SDF launches) and hm_sdf.hip (persistent march kernel: a workgroup carries its own rays through ALL rounds).
// Included inside each file's anonymous namespace; both files are compiled with -ffp-contract=off.
//
// Arithmetic is kept bit-compatible with the reference's elementwise fp32 expressions (model/ray_tracing.py:98-187).
#pragma once

enum : int32_t {  // counters (device int32 array)
    C_ROUND0 = 0,        // [0..63] points appended in march round r
    C_NSAMP = 64,        // rays handed to the sampler
    C_NSAMP_PTS = 65,    // C_NSAMP * n_steps
    C_NSEC = 66,         // rays in the secant refinement
    C_NSEL = 67,         // mask-loss rays for the closest-approach search
    C_NSEL_PTS = 68,
    C_ANY_LIVE = 69,     // some ray was still marching after the first evaluation (=> >= 1 global iteration)
    C_EVALS = 70,        // total SDF point evaluations (statistics)
    C_UNFINISHED = 71,   // rays whose state machine had not finished after the last march round
    C_BIG_PTS = 73,      // points of the one big SDF launch: sampler points followed by closest-approach points
    C_NONFINITE = 72,    // SDF values consumed by the search that were NaN / Inf (must be 0: a NaN fails every
                         // `sdf > threshold` test, so the ray would silently count as converged where it stands)
    C_NSAMP2 = 74,       // lazy sampler: rays whose first sign change is not among the head samples
    C_HEAD_PTS = 75,     // points of the sampler's first pass (= where the closest-approach values start)
    C_TAIL_PTS = 76,     // lazy sampler: points of the second pass
    C_TILE_CURSOR = 77,  // sdf_scan_secant_kernel: next 64-point tile of the closest-approach scan (zero at launch)
    C_COUNT = 80
};

enum : uint8_t { ST_WAIT_FIRST = 1, ST_WAIT_MARCH = 2, ST_WAIT_LS = 3, ST_DONE = 4 };

struct TraceWs {
    float *t_s, *t_e, *t_min, *t_max, *cur_s, *cur_e, *nxt_s, *nxt_e;  // [N]
    float *z_lo, *z_hi, *v_lo, *v_hi, *z;                               // [N] secant state by secant slot
    int32_t *slot_s, *slot_e, *list_samp, *list_sec, *list_sel;          // [N]
    int32_t *tail_slot;      // [N] by sampler slot: index in the second pass, -1 = resolved by the head samples
    uint8_t *live_s, *live_e, *stage, *it, *k, *is_samp;                // [N]
    float *pts;    // [cap,3]
    float *vals;   // [cap]
    int32_t *cnt;  // [C_COUNT]
    int64_t cap;
};

struct TraceArgs {
    TraceWs w;
    const float *cam;        // [B,3]
    const float *dirs;       // [N,3]
    const uint8_t *obj;      // [N]
    const float *t_sphere;   // [N,2]
    const uint8_t *hit;      // [N]
    const float *fracs;      // [n_steps] linspace(0,1)
    const float *steps_u;    // [n_steps] shared random fractions (training tail)
    float *out_pts;          // [N,3]
    uint8_t *out_mask;       // [N]
    float *out_t;            // [N]
    int64_t n, rays_per_image;
    float thr;
    float back[4];           // (1 - line_search_step) / 2^k, formed in double on the host
    int32_t ls_iters, max_it, n_steps, n_secant, training;
    int32_t head;            // lazy sampler: samples 0..head-1 and n_steps-1 form the first pass (0 = all in one pass)
    int64_t tail_off;        // lazy sampler: where the second pass starts in pts / vals (host-known: the buffers are
                             // sized for 3 * N * n_steps points)
    int64_t sel_off;         // closest-approach points: their own (third) region of pts / vals when they are scanned by a
                             // launch of their own (hm_trace_scan_secant); -1: behind the sampler's first pass
};

__device__ __forceinline__ void along(const TraceArgs &a, int64_t i, float t, float &px, float &py, float &pz) {
    const float *c = a.cam + (i / a.rays_per_image) * 3;
    const float *d = a.dirs + i * 3;
    px = __fadd_rn(c[0], __fmul_rn(t, d[0]));
    py = __fadd_rn(c[1], __fmul_rn(t, d[1]));
    pz = __fadd_rn(c[2], __fmul_rn(t, d[2]));
}

// `cursor` counts the points of the round; the point's slot in pts / vals is slot_base + its ticket.  The launch-per-round
// search uses a global cursor (slot_base 0: one compact list for the whole batch); the persistent tail of the march
// (hm_sdf.hip: trace_march_tail_kernel) gives every workgroup a cursor in LDS and 16 slots of its own.
__device__ __forceinline__ int32_t append_point(const TraceArgs &a, int32_t *cursor, int32_t slot_base, int64_t i,
                                                float t) {
    const int32_t idx = slot_base + atomicAdd(cursor, 1);
    float px, py, pz;
    along(a, i, t, px, py, pz);
    a.w.pts[(int64_t)idx * 3 + 0] = px;
    a.w.pts[(int64_t)idx * 3 + 1] = py;
    a.w.pts[(int64_t)idx * 3 + 2] = pz;
    return idx;
}

// round 0: both sphere intersections of every ray that hits the bounding sphere (ray_tracing.py:101-128)
__device__ __forceinline__ void trace_init_ray(const TraceArgs &a, int64_t i, int32_t *cursor, int32_t slot_base) {
    const TraceWs &w = a.w;
    const bool hit = a.hit[i] != 0;
    const float ts = hit ? a.t_sphere[2 * i] : 0.0f, te = hit ? a.t_sphere[2 * i + 1] : 0.0f;
    w.t_s[i] = ts; w.t_e[i] = te; w.t_min[i] = ts; w.t_max[i] = te;
    w.cur_s[i] = 0.0f; w.cur_e[i] = 0.0f; w.nxt_s[i] = 0.0f; w.nxt_e[i] = 0.0f;
    w.live_s[i] = hit; w.live_e[i] = hit;
    w.it[i] = 0; w.k[i] = 0; w.is_samp[i] = 0;
    w.stage[i] = ST_WAIT_FIRST;
    w.slot_s[i] = hit ? append_point(a, cursor, slot_base, i, ts) : -1;
    w.slot_e[i] = hit ? append_point(a, cursor, slot_base, i, te) : -1;
}

// one round of the per-ray sphere-tracing state machine (ray_tracing.py:130-186)
// cursor / slot_base: where the points of the NEXT round go (see append_point)
__device__ __forceinline__ void trace_advance_ray(const TraceArgs &a, int64_t i, int32_t *cursor, int32_t slot_base) {
    const TraceWs &w = a.w;
    uint8_t st = w.stage[i];
    if (st == ST_DONE) return;
    float t_s = w.t_s[i], t_e = w.t_e[i], cur_s = w.cur_s[i], cur_e = w.cur_e[i];
    float nxt_s = w.nxt_s[i], nxt_e = w.nxt_e[i];
    bool live_s = w.live_s[i], live_e = w.live_e[i];
    int it = w.it[i], k = w.k[i];
    const int32_t ss = w.slot_s[i], se = w.slot_e[i];
    if (ss >= 0) nxt_s = w.vals[ss];
    if (se >= 0) nxt_e = w.vals[se];
    if ((ss >= 0 && !isfinite(nxt_s)) || (se >= 0 && !isfinite(nxt_e))) atomicAdd(w.cnt + C_NONFINITE, 1);
    int32_t new_ss = -1, new_se = -1;
    const bool first = (st == ST_WAIT_FIRST);
    bool over_s = false, over_e = false;
    // a ray moves through at most: LS-result -> LS-check -> top-of-loop -> march, i.e. 2 passes
    for (int pass = 0; pass < 3; ++pass) {
        if (st == ST_WAIT_FIRST) {
            // top of the reference's while-loop: threshold, update masks, maybe stop, else march
            cur_s = live_s ? nxt_s : 0.0f;
            if (cur_s <= a.thr) cur_s = 0.0f;
            cur_e = live_e ? nxt_e : 0.0f;
            if (cur_e <= a.thr) cur_e = 0.0f;
            live_s = live_s && (cur_s > a.thr);
            live_e = live_e && (cur_e > a.thr);
            if (first && pass == 0 && (live_s || live_e)) atomicOr(w.cnt + C_ANY_LIVE, 1);
            if (!(live_s || live_e) || it == a.max_it) {
                st = ST_DONE;
                break;
            }
            ++it;
            t_s = __fadd_rn(t_s, cur_s);
            t_e = __fsub_rn(t_e, cur_e);
            nxt_s = 0.0f;
            nxt_e = 0.0f;
            if (live_s) new_ss = append_point(a, cursor, slot_base, i, t_s);
            if (live_e) new_se = append_point(a, cursor, slot_base, i, t_e);
            st = ST_WAIT_MARCH;
            break;
        }
        if (st == ST_WAIT_MARCH) {
            k = 0;
        } else {  // ST_WAIT_LS
            ++k;
        }
        over_s = nxt_s < 0.0f;
        over_e = nxt_e < 0.0f;
        if (k < a.ls_iters && (over_s || over_e)) {
            // pull a step that landed inside the surface back by (1 - step)/2^k of the last move
            const float back = a.back[k];
            if (over_s) {
                t_s = __fsub_rn(t_s, __fmul_rn(back, cur_s));
                new_ss = append_point(a, cursor, slot_base, i, t_s);
            }
            if (over_e) {
                t_e = __fadd_rn(t_e, __fmul_rn(back, cur_e));
                new_se = append_point(a, cursor, slot_base, i, t_e);
            }
            st = ST_WAIT_LS;
            break;
        }
        const bool crossed = t_s < t_e;
        live_s = live_s && crossed;
        live_e = live_e && crossed;
        st = ST_WAIT_FIRST;  // fall through to the top of the loop with the values we already hold
    }
    w.stage[i] = st;
    w.t_s[i] = t_s; w.t_e[i] = t_e; w.cur_s[i] = cur_s; w.cur_e[i] = cur_e;
    w.nxt_s[i] = nxt_s; w.nxt_e[i] = nxt_e;
    w.live_s[i] = live_s; w.live_e[i] = live_e;
    w.it[i] = (uint8_t)it; w.k[i] = (uint8_t)k;
    w.slot_s[i] = new_ss; w.slot_e[i] = new_se;
}

__device__ __forceinline__ float secant_z(float v_lo, float v_hi, float z_lo, float z_hi) {
    // - sdf_low * (z_high - z_low) / (sdf_high - sdf_low) + z_low, evaluated left to right
    return __fadd_rn(__fdiv_rn(__fmul_rn(-v_lo, __fsub_rn(z_hi, z_lo)), __fsub_rn(v_hi, v_lo)), z_lo);
}


// one secant iteration of secant ray q (ray_tracing.py:255-266); the last one writes the refined hit
__device__ __forceinline__ void secant_advance_ray(const TraceArgs &a, int64_t q, int last) {
    const TraceWs &w = a.w;
    const float v = w.vals[q];
    if (!isfinite(v)) atomicAdd(w.cnt + C_NONFINITE, 1);
    float z = w.z[q], z_lo = w.z_lo[q], z_hi = w.z_hi[q], v_lo = w.v_lo[q], v_hi = w.v_hi[q];
    if (v > 0.0f) { z_lo = z; v_lo = v; }
    if (v < 0.0f) { z_hi = z; v_hi = v; }
    z = secant_z(v_lo, v_hi, z_lo, z_hi);
    w.z[q] = z; w.z_lo[q] = z_lo; w.z_hi[q] = z_hi; w.v_lo[q] = v_lo; w.v_hi[q] = v_hi;
    const int64_t i = w.list_sec[q];
    float px, py, pz;
    along(a, i, z, px, py, pz);
    if (last) {
        a.out_t[i] = z;
        w.t_s[i] = z;
        a.out_pts[i * 3] = px; a.out_pts[i * 3 + 1] = py; a.out_pts[i * 3 + 2] = pz;
    } else {
        w.pts[q * 3] = px; w.pts[q * 3 + 1] = py; w.pts[q * 3 + 2] = pz;
    }
}


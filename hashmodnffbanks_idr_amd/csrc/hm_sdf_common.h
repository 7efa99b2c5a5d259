// hm_sdf_common.h - device helpers shared by the fused SDF kernels (hm_sdf.hip: exact fp32; hm_sdf_bf16.hip: bf16
// coarse-search variant).  Included inside each file's anonymous namespace.
#pragma once
typedef float f32x16 __attribute__((ext_vector_type(16)));


struct SdfNet {  // by value -> kernarg
    int32_t n_layers;
    int32_t x_groups;    // k-groups (of 4) in the X region
    int32_t emb_groups;  // k-groups in the EMB region
    float beta;          // Laplace density beta = |beta_param| + beta_min
    int64_t emb_stride;  // > 0: `x` holds PRECOMPUTED embedding rows (emb_stride floats apart, lv.E used) - the encode
                         //      phase becomes a tile load (embedders other than the plain hash grid, hm_sdf_fwd_emb)
    hm_mlp_layer layer[HM_MAX_LAYERS];
};

// 16 bytes of a packed weight image: buffer load with the descriptor in SGPRs, `voff` = lane * 16 (one loop-invariant
// VGPR) and the tile / octet offset as the SCALAR offset - no per-load 64-bit address arithmetic on the VALU, whose
// instructions are serial with the MFMAs of both waves on a SIMD (r3af: 128 -> 132 TFLOP/s on the 64-point kernel)
__device__ __forceinline__ float4 ld_w16(const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) {
    const auto u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
}

// tile of precomputed embedding rows -> EMB[(e/4)][p][e%4] (group stride gf floats), zero padded to 4*egroups columns
__device__ __forceinline__ void load_emb_tile(float *EMB, const float *__restrict__ emb, int64_t stride, int64_t base,
                                              int cnt, int E, int egroups, int pts, int gf, int tid, int nthreads) {
    const int epad = egroups * 4;
    for (int i = tid; i < pts * epad; i += nthreads) {
        const int p = i / epad, e = i - p * epad;
        EMB[(e >> 2) * gf + p * 4 + (e & 3)] = (p < cnt && e < E) ? emb[(base + p) * stride + e] : 0.0f;
    }
}

// nn.Softplus(beta=100, threshold=20): y = x if 100x > 20 else log1p(exp(100x))/100
// evaluated as max(a,0) + ln(1 + 2^(-|z| log2 e)) / 100, z = 100 a, with the native exp2/log2 units (abs error of the
// logarithm term < 1e-9 after the scaling): 9 VALU instructions.  The 1/100 is folded into the log2 -> ln constant: an
// IEEE fp32 division here was a 10-instruction v_div_scale / v_rcp / v_fma / v_div_fixup sequence per activation -
// more than the rest of the function - and the epilogue VALU work is serial with the MFMAs on gfx950.
__device__ __forceinline__ float softplus100(float a) {
    const float z = a * 100.0f;
    const float t = __builtin_amdgcn_exp2f(-fabsf(z) * 1.4426950408889634f);
    const float l = __builtin_amdgcn_logf(1.0f + t) * 0.006931471805599453f;  // v_log_f32 is log2; ln 2 / 100
    const float sp = fmaxf(a, 0.0f) + l;
    return z > 20.0f ? a : sp;
}

// Four activations at once on the packed fp32 VALU ops (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two lanes' worth per
// issue slot): 6.5 instead of 11 instructions per activation.  The epilogue is VALU work that cannot overlap the MFMAs
// (in-place activations, workgroup barriers): in the split-operand kernel it costs as many SIMD cycles as the matrix
// products themselves (profiles/r03_split_mfma_pmc.json), in the fp32 kernel ~19 %.  The threshold select of the scalar
// form is dropped: for 100 a > 20 the logarithm term is < 2.1e-11 < half an ulp of a >= 0.2, so max(a, 0) + l == a bit
// for bit - the value nn.Softplus(100, threshold=20) returns.
typedef float f32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2p softplus100_x2(f32x2p a) {
    const f32x2p na = {-fabsf(a.x), -fabsf(a.y)};
    const f32x2p arg = na * 144.26950408889634f;                       // -|100 a| log2(e)
    f32x2p t = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    t = t + 1.0f;
    const f32x2p lg = {__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
    const f32x2p mx = {fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f)};
    return __builtin_elementwise_fma(lg, (f32x2p){0.006931471805599453f, 0.006931471805599453f}, mx);   // ln 2 / 100
}
__device__ __forceinline__ void softplus100_4(float &v0, float &v1, float &v2, float &v3) {
    const f32x2p a = softplus100_x2((f32x2p){v0, v1}), b = softplus100_x2((f32x2p){v2, v3});
    v0 = a.x; v1 = a.y; v2 = b.x; v3 = b.y;
}

// density_net.py:20-30 + implicit_differentiable_renderer.py:112
__device__ __forceinline__ float sdf_clamp(float s, float beta) {
    const float alpha = 1.0f / beta;
    const float sg = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f);
    const float rho = alpha * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
    return tanhf(s / (2.0f + rho));
}

template <int FRAC>
__device__ __forceinline__ void corner(float x, int32_t res, int bit, uint32_t &u, float &w) {
    const float xs = __fmul_rn(x, (float)res);
    if (FRAC == HM_FRAC_REFERENCE) {
        u = (uint32_t)((int32_t)xs) + (uint32_t)bit;
        w = bit ? 0.0f : 1.0f;
    } else {
        const float fl = floorf(xs);
        const float xf = __fsub_rn(xs, fl);
        u = (uint32_t)((int32_t)fl) + (uint32_t)bit;
        w = bit ? xf : __fsub_rn(1.0f, xf);
    }
}


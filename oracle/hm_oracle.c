/*
 * hm_oracle.c - CPU restatement (plain C) of the reference's per-ray hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (hashmodnffbanks_idr_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors produced
 * by importing the reference's own Python in the build container
 * (tests/golden/make_goldens.py -> tests/golden/ *.npz; tests/test_oracle_golden.py).
 *
 * Reference citations are relative to /root/reference/code/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define HMO_API __attribute__((visibility("default")))

/* model/embeddings/hashGridEmbedding.py:14 - only the first three primes are used for 3-D input */
static const uint32_t HMO_PRIMES[3] = {1u, 3u, 2654435761u};

/* ------------------------------------------------------------------------------------ */
/* hashGridEmbedding.py:126-132: beta growth, floor(base*beta^l), rows=min(res^dim, 2^T) */
HMO_API int hmo_level_table(int n_levels, int log2_hashmap_size, int base_resolution, int desired_resolution,
                            int in_dim, int32_t *res, uint32_t *rows, uint64_t *row_off) {
    if (n_levels < 2 || in_dim != 3) return -1;
    double beta = exp((log((double)desired_resolution) - log((double)base_resolution)) / (double)(n_levels - 1));
    uint64_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        double r = floor((double)base_resolution * pow(beta, (double)l));
        int64_t ri = (int64_t)r;
        uint64_t cube = (uint64_t)ri * (uint64_t)ri * (uint64_t)ri;
        uint64_t cap = 1ull << log2_hashmap_size;
        res[l] = (int32_t)ri;
        rows[l] = (uint32_t)(cube < cap ? cube : cap);
        row_off[l] = off;
        off += rows[l];
    }
    row_off[n_levels] = off;
    return 0;
}

/* hashGridEmbedding.py:84-85: xs = x*res (fp32), xi = trunc toward zero (.long()) */
static inline int32_t hmo_trunc_voxel(float x, int32_t res) {
    volatile float xs = x * (float)res; /* volatile: keep the fp32 rounding of the product */
    return (int32_t)xs;                 /* C cast truncates toward zero, like Tensor.long() */
}

/* hashGridEmbedding.py:32-40: ((idx*prime) & 0xffffffff) xor-folded, then % hashmap_size.
 * The low 32 bits of an int64 product depend only on the low 32 bits of the factors, so a
 * wrapping uint32 multiply is the same function. */
static inline uint32_t hmo_hash3(uint32_t ux, uint32_t uy, uint32_t uz, uint32_t rows) {
    uint32_t h = (ux * HMO_PRIMES[0]) ^ (uy * HMO_PRIMES[1]) ^ (uz * HMO_PRIMES[2]);
    return h % rows;
}

/* corner n uses xi[d]+1 where bit d of n is set (bin_mask, hashGridEmbedding.py:76-79,93) */
HMO_API void hmo_corner_ids(const float *x, int64_t n, int32_t res, uint32_t rows, int32_t *xi_out,
                            uint32_t *ids_out) {
    for (int64_t i = 0; i < n; ++i) {
        int32_t xi[3];
        for (int d = 0; d < 3; ++d) {
            xi[d] = hmo_trunc_voxel(x[3 * i + d], res);
            if (xi_out) xi_out[3 * i + d] = xi[d];
        }
        for (int c = 0; c < 8; ++c) {
            uint32_t u[3];
            for (int d = 0; d < 3; ++d) u[d] = (uint32_t)xi[d] + (uint32_t)((c >> d) & 1);
            ids_out[8 * i + c] = hmo_hash3(u[0], u[1], u[2], rows);
        }
    }
}

/* Encoder output layout (hashGridEmbedding.py:150-155, frequency_enc.py:63-67):
 *   [ x(3) | sin(2*pi*x@B)(L) | cos(..)(L) | level0 f0..fF-1 | ... | level L-1 ]          */
/* frac_mode 0 = "reference": xf = x - x.float() == 0 (hashGridEmbedding.py:86), so corner 0
 *               has weight 1 and the other seven weight 0; all 8 rows are still gathered and
 *               multiplied (hashGridEmbedding.py:93-102).
 * frac_mode 1 = "trilinear" (build-defined, NOT a parity mode): floor + fractional weights.  */
HMO_API void hmo_encode_fwd(int L, int F, const int32_t *res, const uint32_t *rows, const uint64_t *row_off,
                            const float *x, int64_t n, const float *table, const float *Bf, float *out,
                            int frac_mode) {
    const int E = 3 + 2 * L + L * F;
    const float two_pi = (float)(2.0 * 3.14159265358979323846); /* 2*np.pi*x is evaluated in fp32 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *p = x + 3 * i;
        float *o = out + (int64_t)E * i;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        volatile float s0 = two_pi * p[0], s1 = two_pi * p[1], s2 = two_pi * p[2];
        for (int c = 0; c < L; ++c) {
            /* matmul [N,3]@[3,L] in fp32: torch's CPU sgemm evaluates it as a k-ordered fma chain
             * (measured against the reference output: this form reproduces the argument bit-for-bit) */
            float acc = s0 * Bf[0 * L + c];
            acc = fmaf(s1, Bf[1 * L + c], acc);
            acc = fmaf(s2, Bf[2 * L + c], acc);
            o[3 + c] = sinf(acc);
            o[3 + L + c] = cosf(acc);
        }
        for (int l = 0; l < L; ++l) {
            float w3[3][2];
            int32_t xi[3];
            for (int d = 0; d < 3; ++d) {
                volatile float xs = p[d] * (float)res[l];
                if (frac_mode == 0) {
                    xi[d] = (int32_t)xs;
                    w3[d][0] = 1.0f; w3[d][1] = 0.0f;
                } else {
                    float fl = floorf(xs);
                    xi[d] = (int32_t)fl;
                    float xf = xs - fl;
                    w3[d][0] = 1.0f - xf; w3[d][1] = xf;
                }
            }
            float acc[8];
            for (int f = 0; f < F; ++f) acc[f] = 0.0f;
            for (int c = 0; c < 8; ++c) {
                uint32_t u[3];
                float w = 1.0f;
                for (int d = 0; d < 3; ++d) {
                    int b = (c >> d) & 1;
                    u[d] = (uint32_t)xi[d] + (uint32_t)b;
                    w = w * w3[d][b];
                }
                uint32_t id = hmo_hash3(u[0], u[1], u[2], rows[l]);
                const float *row = table + (row_off[l] + id) * (uint64_t)F;
                for (int f = 0; f < F; ++f) acc[f] = acc[f] + row[f] * w;
            }
            for (int f = 0; f < F; ++f) o[3 + 2 * L + l * F + f] = acc[f];
        }
    }
}

/* Table gradient: d_table[row] += w * d_out (embedding_dense_backward of hashGridEmbedding.py:99-102).
 * d_table must be zeroed by the caller.  Serial so the summation order is deterministic. */
HMO_API void hmo_encode_bwd_table(int L, int F, const int32_t *res, const uint32_t *rows, const uint64_t *row_off,
                                  const float *x, int64_t n, const float *d_out, float *d_table, int frac_mode) {
    const int E = 3 + 2 * L + L * F;
    for (int64_t i = 0; i < n; ++i) {
        const float *p = x + 3 * i;
        const float *g = d_out + (int64_t)E * i + 3 + 2 * L;
        for (int l = 0; l < L; ++l) {
            float w3[3][2];
            int32_t xi[3];
            for (int d = 0; d < 3; ++d) {
                volatile float xs = p[d] * (float)res[l];
                if (frac_mode == 0) {
                    xi[d] = (int32_t)xs; w3[d][0] = 1.0f; w3[d][1] = 0.0f;
                } else {
                    float fl = floorf(xs);
                    xi[d] = (int32_t)fl;
                    float xf = xs - fl;
                    w3[d][0] = 1.0f - xf; w3[d][1] = xf;
                }
            }
            for (int c = 0; c < 8; ++c) {
                uint32_t u[3];
                float w = 1.0f;
                for (int d = 0; d < 3; ++d) {
                    int b = (c >> d) & 1;
                    u[d] = (uint32_t)xi[d] + (uint32_t)b;
                    w = w * w3[d][b];
                }
                uint32_t id = hmo_hash3(u[0], u[1], u[2], rows[l]);
                float *row = d_table + (row_off[l] + id) * (uint64_t)F;
                for (int f = 0; f < F; ++f) row[f] += w * g[l * F + f];
            }
        }
    }
}

/* nn.utils.weight_norm, dim=0: W = g * v / ||v||_row (implicit_differentiable_renderer.py:80-81) */
HMO_API void hmo_fold_weight_norm(const float *v, const float *g, int out_dim, int in_dim, float *w) {
    for (int o = 0; o < out_dim; ++o) {
        double s = 0.0;
        for (int k = 0; k < in_dim; ++k) s += (double)v[(int64_t)o * in_dim + k] * (double)v[(int64_t)o * in_dim + k];
        float nrm = (float)sqrt(s);
        float sc = g[o] / nrm;
        for (int k = 0; k < in_dim; ++k) w[(int64_t)o * in_dim + k] = v[(int64_t)o * in_dim + k] * sc;
    }
}

/* nn.Softplus(beta=100), default threshold 20 (implicit_differentiable_renderer.py:84) */
static inline float hmo_softplus100(float a) {
    float z = a * 100.0f;
    return z > 20.0f ? a : log1pf(expf(z)) / 100.0f;
}

/* density_net.py:20-30 (beta = |0.9| + 1e-4, evaluated under no_grad) and
 * implicit_differentiable_renderer.py:112: sdf = tanh(s / (2 + rho(s)))                     */
static inline float hmo_sdf_clamp(float s, float beta) {
    float alpha = 1.0f / beta;
    float sg = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f);
    float rho = alpha * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
    return tanhf(s / (2.0f + rho));
}

/*
 * SDF network forward (implicit_differentiable_renderer.py:89-113) on an already-embedded input.
 *   emb [n,E]; layer l: W_l [out_l,in_l] row-major (weight-norm already folded), b_l [out_l];
 *   skip_layer: layer index whose input is cat[x, emb]/sqrt(2) (or -1).
 *   out [n, out_last]; column 0 gets the tanh/Laplace clamp.
 */
HMO_API int hmo_mlp_fwd(int n_layers, const int32_t *in_dims, const int32_t *out_dims, const float *const *W,
                        const float *const *b, int skip_layer, const float *emb, int E, int64_t n, float beta,
                        int apply_clamp, float *out) {
    int maxw = E;
    for (int l = 0; l < n_layers; ++l) {
        if (in_dims[l] > maxw) maxw = in_dims[l];
        if (out_dims[l] > maxw) maxw = out_dims[l];
    }
    const float sqrt2 = (float)sqrt(2.0);
    int bad = 0;
#pragma omp parallel
    {
        float *a = (float *)malloc(sizeof(float) * (size_t)maxw);
        float *c = (float *)malloc(sizeof(float) * (size_t)maxw);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const float *e = emb + (int64_t)E * i;
            int cur = E;
            memcpy(a, e, sizeof(float) * (size_t)E);
            for (int l = 0; l < n_layers; ++l) {
                if (l == skip_layer) {
                    for (int k = 0; k < E; ++k) a[cur + k] = e[k];
                    cur += E;
                    for (int k = 0; k < cur; ++k) a[k] = a[k] / sqrt2;
                }
                if (cur != in_dims[l]) { bad = 1; continue; }
                const float *Wl = W[l];
                for (int o = 0; o < out_dims[l]; ++o) {
                    const float *wr = Wl + (int64_t)o * cur;
                    float acc = 0.0f;
                    for (int k = 0; k < cur; ++k) acc += a[k] * wr[k];
                    acc += b[l][o];
                    c[o] = (l < n_layers - 1) ? hmo_softplus100(acc) : acc;
                }
                cur = out_dims[l];
                float *t = a; a = c; c = t;
            }
            float *o = out + (int64_t)cur * i;
            memcpy(o, a, sizeof(float) * (size_t)cur);
            if (apply_clamp) o[0] = hmo_sdf_clamp(o[0], beta);
        }
        free(a);
        free(c);
    }
    return bad ? -1 : 0;
}

/* utils/rend_util.py:141-162: near/far intersection of rays with the sphere |p|=r (one camera). */
HMO_API void hmo_sphere_intersection(const float *cam, const float *dirs, int64_t n, float r, float *t2,
                                     uint8_t *mask) {
    float cc = cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2];
    float nrm = sqrtf(cc);
    for (int64_t i = 0; i < n; ++i) {
        const float *d = dirs + 3 * i;
        float dc = d[0] * cam[0] + d[1] * cam[1] + d[2] * cam[2];
        float us = dc * dc - (nrm * nrm - r * r);
        mask[i] = us > 0.0f;
        if (mask[i]) {
            float sq = sqrtf(us);
            float a = sq * -1.0f - dc, b = sq * 1.0f - dc;
            t2[2 * i] = a < 0.0f ? 0.0f : a;
            t2[2 * i + 1] = b < 0.0f ? 0.0f : b;
        } else {
            t2[2 * i] = 0.0f;
            t2[2 * i + 1] = 0.0f;
        }
    }
}

HMO_API int hmo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

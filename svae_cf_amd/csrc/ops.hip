// HBM-bound kernels of the SVAE step for gfx950: embedding gather / segmented scatter-add, latent
// reparameterisation + KL, row log-softmax + next-k NLL over materialised logits, fused Adam(L2),
// column sums (bias gradients), masked top-k, Philox normal noise, small helpers.
#include <mutex>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// item embedding (reference nn.Embedding call nb:756; backward = dense scatter-add via nb:1205)
// ------------------------------------------------------------------------------------------------
__global__ void embed_fwd_kernel(const float* __restrict__ table, const int* __restrict__ x, float* __restrict__ out,
                                 int R, int I) {
    // one wave per row: 64 lanes x float4 = 256 floats = one embedding row (1 KiB, fully coalesced)
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int stride = gridDim.x * (blockDim.x >> 6);
    for (; row < R; row += stride) {
        int it = x[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it >= 0 && it < I) v = reinterpret_cast<const float4*>(table + (size_t)it * SV_E)[lane];
        reinterpret_cast<float4*>(out + (size_t)row * SV_E)[lane] = v;
    }
}

// Deterministic scatter-add: rows are pre-sorted by item on the host (perm), the rows of one item are summed
// in a fixed order and stored -- no atomics, bitwise reproducible.  Item popularity is heavily skewed (a
// popular item is in every user's history), so segments come sorted by length: the first `nlong` (more
// than EMB_LONG rows) get a whole workgroup each, the rest one wave each; both keep 4 rows in flight per
// wave (one wave walking a 256-row segment one dependent load at a time was the kernel's whole run time).
constexpr int EMB_LONG = 16;

__device__ __forceinline__ float4 embed_seg_sum(const float* __restrict__ demb, const int* __restrict__ perm, int b,
                                                int e, int first, int step, int lane) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = b + first; i < e; i += 4 * step) {
        int r[4];
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (i + j * step < e) ? perm[i + j * step] : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const float4*>(demb + (size_t)(r[j] < 0 ? 0 : r[j]) * SV_E)[lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r[j] >= 0) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    return acc;
}

// One launch for both kinds of segment: workgroups 0..nlong-1 take one long segment each (all four waves, summed
// through LDS in a fixed order), the rest walk the short segments one wave per segment.  Long segments first: they
// are the latency chains (up to B rows of one popular item), and the short ones fill the GPU beside them -- as two
// launches on one stream the second waited for the first (24 + 11 us; 26 us as one).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ demb, const int* __restrict__ perm,
                                                        const int* __restrict__ seg_off,
                                                        const int* __restrict__ seg_item,
                                                        float* __restrict__ gtable, int nlong, int nseg, float scale) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= nlong) {
        int s = nlong + ((int)blockIdx.x - nlong) * 4 + wave;
        const int stride = ((int)gridDim.x - nlong) * 4;
        for (; s < nseg; s += stride) {
            float4 acc = embed_seg_sum(demb, perm, seg_off[s], seg_off[s + 1], 0, 1, lane);
            acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
            reinterpret_cast<float4*>(gtable + (size_t)seg_item[s] * SV_E)[lane] = acc;
        }
        return;
    }
    const int s = blockIdx.x;
    red[wave][lane] = embed_seg_sum(demb, perm, seg_off[s], seg_off[s + 1], wave, 4, lane);
    __syncthreads();
    if (wave != 0) return;
    float4 acc = red[0][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float4 u = red[q][lane];
        acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
    }
    acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
    reinterpret_cast<float4*>(gtable + (size_t)seg_item[s] * SV_E)[lane] = acc;
}

// ------------------------------------------------------------------------------------------------
// latent: mu / log_sigma split, z = mu + exp(log_sigma) * eps, KL row term
// (reference Model.sample_latent nb:734-750, VAELoss KL nb:797).  log_sigma is exponentiated as a
// log-std for sampling but enters KL as a log-variance: reproduced as is.
// ------------------------------------------------------------------------------------------------
__global__ void latent_fwd_kernel(const float* __restrict__ tmp, const float* __restrict__ eps, float* __restrict__ z,
                                  float* __restrict__ kl_row, int R) {
    // one wave per row, lane = latent dim
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int stride = gridDim.x * (blockDim.x >> 6);
    for (; row < R; row += stride) {
        const float mu = tmp[(size_t)row * SV_L2 + lane];
        const float ls = tmp[(size_t)row * SV_L2 + SV_L + lane];
        const float sg = expf(ls);
        z[(size_t)row * SV_L + lane] = mu + sg * eps[(size_t)row * SV_L + lane];
        float kl = 0.5f * (-ls + sg + mu * mu - 1.0f);
        kl = sv_wave_sum(kl);
        if (lane == 0) kl_row[row] = kl;
    }
}

// dtmp[:, :64] = dz + a*w[r]*mu ; dtmp[:, 64:] = dz*sigma*eps + a*w[r]*0.5*(exp(ls)-1),  a = anneal
__global__ void latent_bwd_kernel(const float* __restrict__ tmp, const float* __restrict__ eps,
                                  const float* __restrict__ dz, const float* __restrict__ w_kl, float anneal,
                                  float* __restrict__ dtmp, int R) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int stride = gridDim.x * (blockDim.x >> 6);
    for (; row < R; row += stride) {
        const float mu = tmp[(size_t)row * SV_L2 + lane];
        const float ls = tmp[(size_t)row * SV_L2 + SV_L + lane];
        const float sg = expf(ls);
        const float g = dz[(size_t)row * SV_L + lane];
        const float kw = anneal * w_kl[row];
        dtmp[(size_t)row * SV_L2 + lane] = g + kw * mu;
        dtmp[(size_t)row * SV_L2 + SV_L + lane] = g * sg * eps[(size_t)row * SV_L + lane] + kw * 0.5f * (sg - 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// row log-softmax + next-k multinomial NLL over materialised logits, gradient written in place
// (reference VAELoss.forward nb:802-808).  One workgroup per row; the row is read twice (second
// read is an L2 hit for |items| up to ~1M) and written once.
//   nll_row[r]  = sum_{valid targets} (lse - logit[tgt])
//   dlogits[r,i] = w_nll[r] * (n_r * softmax[r,i] - y[r,i])      (n_r = valid targets in row r)
// Pad columns I..ldI-1 are written as 0.
// ------------------------------------------------------------------------------------------------
template <bool WRITE_GRAD>
__global__ __launch_bounds__(256) void softmax_nll_kernel(float* __restrict__ logits, const int* __restrict__ tgt,
                                                           const float* __restrict__ w_nll, float* __restrict__ nll_row,
                                                           float* __restrict__ lse_out, int I, int ldI, int K) {
    __shared__ float red[8];
    __shared__ float bc[2];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* lr = logits + (size_t)row * ldI;
    const int n4 = I >> 2;
    // pass 1: online max / sum-exp
    float m = -INFINITY, s = 0.f;
    for (int i = tid; i < n4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(lr)[i];
        const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        if (mx > m) { s *= expf(m - mx); m = mx; }
        s += expf(v.x - m) + expf(v.y - m) + expf(v.z - m) + expf(v.w - m);
    }
    for (int i = (n4 << 2) + tid; i < I; i += 256) {
        const float v = lr[i];
        if (v > m) { s *= expf(m - v); m = v; }
        s += expf(v - m);
    }
    const float wm = sv_wave_max(m);
    s *= (m == -INFINITY) ? 0.f : expf(m - wm);
    s = sv_wave_sum(s);
    if (lane == 0) { red[wave] = wm; red[4 + wave] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float S = 0.f;
        for (int w = 0; w < 4; ++w) S += red[4 + w] * expf(red[w] - M);
        bc[0] = M + logf(S);
        if (lse_out) lse_out[row] = bc[0];
    }
    __syncthreads();
    const float lse = bc[0];
    // targets: -1 = padding.  duplicates were removed on the host (the reference's scatter_ of 1.0 makes a
    // duplicate count once, nb:510-513).  Any K: `postfix` rows have up to T - 1 targets (nb:515-519); thread t
    // takes targets t, t + 256, ...; partial sums are combined in a fixed order.
    {
        float nll = 0.f, nr = 0.f;
        for (int k = tid; k < K; k += 256) {
            const int it = tgt[(size_t)row * K + k];
            if (it >= 0) { nll += lse - lr[it]; nr += 1.f; }
        }
        nll = sv_wave_sum(nll);
        nr = sv_wave_sum(nr);
        if (lane == 0) { red[wave] = nll; red[4 + wave] = nr; }
    }
    __syncthreads();
    if (tid == 0) {
        nll_row[row] = (red[0] + red[1]) + (red[2] + red[3]);
        bc[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
    __syncthreads();
    if (!WRITE_GRAD) return;
    const float w = w_nll[row];
    const float c = bc[1] * w;
    for (int i = tid; i < n4; i += 256) {
        float4 v = reinterpret_cast<const float4*>(lr)[i];
        v.x = c * expf(v.x - lse); v.y = c * expf(v.y - lse); v.z = c * expf(v.z - lse); v.w = c * expf(v.w - lse);
        reinterpret_cast<float4*>(lr)[i] = v;
    }
    for (int i = (n4 << 2) + tid; i < ldI; i += 256) lr[i] = i < I ? c * expf(lr[i] - lse) : 0.f;
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        const int it = tgt[(size_t)row * K + k];
        if (it >= 0) lr[it] -= w;
    }
}

// ------------------------------------------------------------------------------------------------
// Adam with L2 weight decay on one flat buffer (reference torch.optim.Adam defaults, nb:1260-1263)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n4, float b1, float b2, float step_size,
                            float inv_sqrt_bc2, float eps, float wd, float gscale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 Mv = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
        float* pp = reinterpret_cast<float*>(&P);
        const float* gg = reinterpret_cast<const float*>(&G);
        float* mm = reinterpret_cast<float*>(&Mv);
        float* vv = reinterpret_cast<float*>(&V);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = gg[c] * gscale + wd * pp[c];
            mm[c] = mm[c] + (gr - mm[c]) * (1.0f - b1);          // lerp, as torch does
            vv[c] = vv[c] * b2 + gr * gr * (1.0f - b2);
            const float denom = sqrtf(vv[c]) * inv_sqrt_bc2 + eps;
            pp[c] = pp[c] - step_size * (mm[c] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = Mv;
        reinterpret_cast<float4*>(v)[i] = V;
    }
}

// ------------------------------------------------------------------------------------------------
// The notebook's other optimizers (nb:1252-1259, nb:1264-1267) on one flat buffer, torch.optim semantics:
// KIND 0 Adagrad   a = state_sum;                 p -= clr * g / (sqrt(a) + eps)
// KIND 1 Adadelta  a = square_avg, b = acc_delta; p -= lr * sqrt(b + eps) / sqrt(a + eps) * g
// KIND 2 RMSprop   a = square_avg;                p -= lr * g / (sqrt(a) + eps)      (no momentum, not centered)
// g = gscale * grad + wd * p; omd = 1 - decay, rounded from the DOUBLE difference as torch does (1.0f - 0.99f is
// 9.3e-7 away from 0.01f, which RMSprop's g / sqrt(a) shows after three steps).  HBM-bound: 16 B read + 8..12 B written per element.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ void opt_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ a,
                           float* __restrict__ b, long long n4, float lr, float decay, float omd, float eps,
                           float wd, float gscale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 A = reinterpret_cast<float4*>(a)[i];
        float4 Bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == 1) Bv = reinterpret_cast<float4*>(b)[i];
        float* pp = reinterpret_cast<float*>(&P);
        const float* gg = reinterpret_cast<const float*>(&G);
        float* aa = reinterpret_cast<float*>(&A);
        float* bb = reinterpret_cast<float*>(&Bv);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = gg[c] * gscale + wd * pp[c];
            if (KIND == 0) {
                aa[c] = aa[c] + gr * gr;
                pp[c] = pp[c] - lr * (gr / (sqrtf(aa[c]) + eps));
            } else if (KIND == 1) {
                aa[c] = aa[c] * decay + gr * gr * omd;
                const float delta = sqrtf(bb[c] + eps) / sqrtf(aa[c] + eps) * gr;
                bb[c] = bb[c] * decay + delta * delta * omd;
                pp[c] = pp[c] - lr * delta;
            } else {
                aa[c] = aa[c] * decay + gr * gr * omd;
                pp[c] = pp[c] - lr * (gr / (sqrtf(aa[c]) + eps));
            }
        }
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(a)[i] = A;
        if (KIND == 1) reinterpret_cast<float4*>(b)[i] = Bv;
    }
}

// ------------------------------------------------------------------------------------------------
// column sums of a [R][ld] matrix (bias gradients), deterministic two-stage
// ------------------------------------------------------------------------------------------------
__global__ void colsum_partial_kernel(const float* __restrict__ a, float* __restrict__ part, int R, int N, int ld,
                                      int rows_per_block) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > R) r1 = R;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += a[(size_t)r * ld + n];
    part[(size_t)blockIdx.y * N + n] = s;
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int nparts) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * N + n];
    out[n] = s;
}

// out[c][r] = in[r][c]
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    int c = blockIdx.x * 32 + threadIdx.x, r = blockIdx.y * 32 + threadIdx.y;
    for (int i = 0; i < 32; i += 8)
        if (c < cols && r + i < rows) tile[threadIdx.y + i][threadIdx.x] = in[(size_t)(r + i) * cols + c];
    __syncthreads();
    int oc = blockIdx.y * 32 + threadIdx.x, orow = blockIdx.x * 32 + threadIdx.y;
    for (int i = 0; i < 32; i += 8)
        if (oc < rows && orow + i < cols) out[(size_t)(orow + i) * rows + oc] = tile[threadIdx.x][threadIdx.y + i];
}

// loss = sum_r ( w_nll[r]*nll_row[r] + anneal*w_kl[r]*kl_row[r] ); single workgroup, fixed order
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ nll_row,
                                                            const float* __restrict__ kl_row,
                                                            const float* __restrict__ w_nll,
                                                            const float* __restrict__ w_kl, float anneal, int R,
                                                            float* __restrict__ out) {
    __shared__ double red[16][2];
    double a = 0.0, b = 0.0;
    for (int r = threadIdx.x; r < R; r += 1024) {
        a += (double)w_nll[r] * (double)nll_row[r];
        b += (double)w_kl[r] * (double)kl_row[r];
    }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double A = 0.0, Bv = 0.0;
        for (int w = 0; w < 16; ++w) { A += red[w][0]; Bv += red[w][1]; }
        out[0] = (float)(A + (double)anneal * Bv);  // total
        out[1] = (float)A;                          // likelihood part
        out[2] = (float)Bv;                         // KL part (un-annealed)
    }
}

// gather rows: out[i][:] = in[idx[i]][:]  (n4 float4 per row)
__global__ void gather_rows_kernel(const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out,
                                   int nrows, int ld_in4, int ld_out4, int n4) {
    const int row = blockIdx.x;
    if (row >= nrows) return;
    const float4* src = reinterpret_cast<const float4*>(in) + (size_t)idx[row] * ld_in4;
    float4* dst = reinterpret_cast<float4*>(out) + (size_t)row * ld_out4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// masked top-k (reference evaluate(): mask nb:621-624, full sort nb:632, [:k] nb:636)
// score_i = s_i - |s_i * m_i| * 1e8, m = one-hot of the user's INPUT items.  One workgroup per user.
// Result = the first K entries of a stable descending sort: ties -> lower item index first.
//
// History: K rounds of arg-max (100 sweeps over the row with two barriers each: 300 us per batch of 256 users at 20k
// items, a hundred times the row's HBM time) -> radix selects + a bitonic sort (rounds 1-2: 17 us) -> round 3:
//   1. the row is read ONCE into registers (16 bytes per lane; catalogues up to 24 576 items -- larger ones stream
//      through registers in chunks, see STREAM), its loads issued before the seen items are set as bits in LDS; every
//      masked score becomes an order-preserving 32-bit key;
//   2. pre-filter bound: every wave finds its ceil(K / 16)-th largest per-thread maximum to 16 bits by a binary search
//      on the key's leading bits (ballot + count; no LDS); the smallest of the 16 results is reached by at least K
//      maxima, so by at least K scores; the ~200 scores that reach it are collected in LDS;
//   3. a candidate's place in the result = how many candidates precede it (larger key, or equal key and lower item):
//      counted by all threads over (candidate, slice of the others) pairs, partial counts met in LDS; places < K are
//      written out.  Four barriers in all.
// Fallbacks, all exact: small catalogues / large K (fewer than 64 maxima per contributing wave) take the bound from a
// radix select over the 1024 maxima; more than 1024 candidates (heavy ties, adversarial layouts) go through a radix
// select for the K-th key + a bitonic sort of the keys above it; K > TOPK_CAP or more than TOPK_CAP ties at the
// threshold fall back to the arg-max rounds (degenerate inputs: e.g. a constant row).
// ------------------------------------------------------------------------------------------------
constexpr int TOPK_CAP = 1024;

__device__ __forceinline__ unsigned topk_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float topk_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// K rounds of arg-max over work[] (destroys it); writes entries k0.. of the result
__device__ void topk_rounds(float* w, int I, int k0, int K, int* out_idx, float* out_val, float* rv, int* ri) {
    const int tid = threadIdx.x;
    for (int k = k0; k < K; ++k) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < I; i += 1024) {
            const float v = w[i];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { rv[tid >> 6] = bv; ri[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = rv[0];
            int fi = ri[0];
            for (int q = 1; q < 16; ++q)
                if (rv[q] > fv || (rv[q] == fv && ri[q] < fi)) { fv = rv[q]; fi = ri[q]; }
            out_idx[k] = fi;
            out_val[k] = fv;
            if (fi >= 0 && fi < I) w[fi] = -INFINITY;
        }
        __syncthreads();
    }
}

// Radix select over n keys (key_at(i), i < n): on return (after a barrier) *s_prefix is the key of the K-th largest
// and *s_remaining how many keys equal to it belong to the K largest.  All 1024 threads call it.
// last_pass > 0 stops after the leading bytes: *s_prefix is then the K-th largest key with its low 8 * last_pass bits
// cleared -- a LOWER BOUND of it, all a pre-filter needs (half the passes and barriers for a few more candidates).
template <class KeyAt>
__device__ __forceinline__ void topk_radix_select(KeyAt key_at, int n, int K, int* hist, unsigned* s_prefix,
                                                  int* s_remaining, int last_pass = 0) {
    const int tid = threadIdx.x;
    if (tid == 0) { *s_prefix = 0u; *s_remaining = K; }
    __syncthreads();
    for (int p = 3; p >= last_pass; --p) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = *s_prefix;
        const unsigned hi_mask = p == 3 ? 0u : (0xFFFFFFFFu << (8 * (p + 1)));
        for (int i = tid; i < n; i += 1024) {
            const unsigned k = key_at(i);
            if ((k & hi_mask) == prefix) atomicAdd(&hist[(k >> (8 * p)) & 255], 1);
        }
        __syncthreads();
        if (tid < 64) {
            // wave 0: lane l owns bins 4l..4l+3; suffix sums over the lanes find the bin that holds the rem-th largest
            // (a serial scan of the 256 bins by one thread took a third of the kernel)
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const int c = h0 + h1 + h2 + h3;
            int suf = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_down(suf, o, 64);
                if (tid + o < 64) suf += v;
            }
            const int rem = *s_remaining;
            const unsigned long long m = __ballot(suf >= rem);      // non-zero: the bins hold at least rem keys
            const int L = 63 - __clzll((long long)m);
            if (tid == L) {
                int r = rem - (suf - c), b;                         // r-th from the top inside my four bins
                if (h3 >= r) b = 3;
                else {
                    r -= h3;
                    if (h2 >= r) b = 2;
                    else {
                        r -= h2;
                        if (h1 >= r) b = 1;
                        else { r -= h1; b = 0; }
                    }
                }
                *s_remaining = r;
                *s_prefix = prefix | ((unsigned)(4 * tid + b) << (8 * p));
            }
        }
        __syncthreads();
    }
}

constexpr int TOPK_NCAND = 4096;
constexpr int TOPK_RV4 = 6;                // float4 per thread of the register-resident row: 24 576 items
constexpr int TOPK_MASKW = 4096;           // 131 072 items

// STREAM: the instantiation for catalogues beyond 24 576 items (the streamed path below; 82 registers, one workgroup per
// CU); the other one keeps the 50 registers that let two workgroups share a CU.
template <bool STREAM>
__global__ __launch_bounds__(1024) void masked_topk_kernel(const float* __restrict__ scores, int ld,
                                                            const int* __restrict__ seen, const int* __restrict__ seen_off,
                                                            int I, int K, int* __restrict__ out_idx,
                                                            float* __restrict__ out_val, float* __restrict__ work) {
    __shared__ float rv[16];
    __shared__ int ri[16];
    __shared__ int hist[256];
    __shared__ unsigned tk[1024];              // per-thread maxima
    __shared__ __attribute__((aligned(16))) unsigned candk[TOPK_NCAND];     // pre-filtered candidates
    __shared__ __attribute__((aligned(16))) int candi[TOPK_NCAND];
    __shared__ unsigned ckey[TOPK_CAP];
    __shared__ int cidx[TOPK_CAP];
    static_assert(TOPK_CAP == 1024, "cidx doubles as one rank counter per thread");
    __shared__ int tidx[TOPK_CAP];
    __shared__ unsigned s_prefix;
    __shared__ int s_remaining, s_ngt, s_nties, s_ncand;
    __shared__ unsigned maskw[TOPK_MASKW];     // seen-item bits of the row (catalogues up to 32 * TOPK_MASKW items)
    const int u = blockIdx.x, tid = threadIdx.x;
    float* w = work + (size_t)u * ld;
    const float* s = scores + (size_t)u * ld;
    int* oi = out_idx + (size_t)u * K;
    float* ov = out_val + (size_t)u * K;
    // Three ways to the candidate list, by catalogue size (all uniform decisions):
    //   in_regs   I <= 24 576: the row is read ONCE, 16 bytes per lane, and stays in registers as sort keys; the seen
    //             items are a bit set in LDS;
    //   streamed  larger catalogues (round 3): the row is still read ONCE, in register-resident chunks of 24 576 items
    //             with the next chunk's loads in flight; the seen items are a small hash set in LDS that is only asked
    //             about the few scores that reach the running bound (masking can only LOWER a score), the bound is
    //             tightened by selecting over the candidate list whenever that fills up;
    //   legacy    anything else (unaligned rows, more than 1024 - K seen items, a flood of ties): the masked row through
    //             the LDS bit set (up to 131 072 items) or a copy in `work`, read twice with 4-byte loads.
    const int nseen = seen_off[u + 1] - seen_off[u];
    const bool aligned = (ld & 3) == 0 && ((uintptr_t)scores & 15) == 0;
    // (the STREAM instantiation is only launched beyond 4096 * TOPK_RV4 items: no register-resident row there)
    const bool in_regs = !STREAM && K <= TOPK_CAP && I <= 32 * TOPK_MASKW && I <= 4096 * TOPK_RV4 && aligned;
    bool streamed = STREAM && K <= TOPK_CAP && !in_regs && aligned && I >= 4096 * TOPK_RV4 && nseen >= 0 && K + nseen <= 1024;
    bool lds_mask = I <= 32 * TOPK_MASKW;
    int* const hset = reinterpret_cast<int*>(maskw);      // streamed: 2048-slot open-addressing set of the seen items
    auto is_seen = [&](int it) -> bool {
        unsigned h = ((unsigned)it * 2654435761u) >> 21;
        for (;;) {
            const int e = hset[h & 2047];
            if (e == it) return true;
            if (e < 0) return false;
            ++h;
        }
    };
    // (legacy / in_regs) the masked row: with the seen items as a bit set in LDS it is computed on the fly from the
    // scores (no masked copy in HBM); catalogues beyond 131 072 items keep the copy in `work`.
    auto mask_setup = [&]() {
        if (lds_mask) {
            for (int i = tid; i < (I + 31) / 32; i += 1024) maskw[i] = 0u;
            __syncthreads();
            for (int i = seen_off[u] + tid; i < seen_off[u + 1]; i += 1024) {
                const int it = seen[i];
                if (it >= 0 && it < I) atomicOr(&maskw[it >> 5], 1u << (it & 31));
            }
        } else {
            for (int i = tid; i < I; i += 1024) w[i] = s[i];
            __syncthreads();
            for (int i = seen_off[u] + tid; i < seen_off[u + 1]; i += 1024) {
                const int it = seen[i];
                if (it >= 0 && it < I) { const float v = s[it]; w[it] = v - fabsf(v * 1.0f) * 100000000.0f; }
            }
        }
        __syncthreads();
    };
    // in_regs: the row's loads are issued before the bit set is built (their HBM latency runs beside its two barriers)
    float4 rowv[TOPK_RV4];
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < TOPK_RV4; ++j) {
            const int i4 = 4 * (tid + 1024 * j);
            rowv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i4 < I) rowv[j] = *reinterpret_cast<const float4*>(s + i4);   // (ld % 4 == 0: the 16 bytes are in the row)
        }
    }
    if (tid == 0) { s_ngt = 0; s_nties = 0; s_ncand = 0; }      // (visible behind the barriers of the set-up below)
    cidx[tid] = 0;                                               // the rank counters of the usual path (TOPK_CAP = 1024)
    if (streamed) {
        for (int i = tid; i < 2048; i += 1024) hset[i] = -1;
        __syncthreads();
        for (int i = seen_off[u] + tid; i < seen_off[u + 1]; i += 1024) {
            const int it = seen[i];
            if (it >= 0 && it < I) {
                unsigned h = ((unsigned)it * 2654435761u) >> 21;
                for (;;) {
                    const int old = atomicCAS(&hset[h & 2047], -1, it);
                    if (old == -1 || old == it) break;
                    ++h;
                }
            }
        }
        __syncthreads();
    } else {
        mask_setup();
    }
    auto val = [&](int i) -> float {
        if (streamed) { const float v = s[i]; return is_seen(i) ? v - fabsf(v * 1.0f) * 100000000.0f : v; }
        if (!lds_mask) return w[i];
        const float v = s[i];
        return ((maskw[i >> 5] >> (i & 31)) & 1u) ? v - fabsf(v * 1.0f) * 100000000.0f : v;
    };
    // the rare paths below (K beyond the cap, a flood of ties) work on the materialised row
    auto materialise = [&]() {
        if (lds_mask || streamed) {
            for (int i = tid; i < I; i += 1024) w[i] = val(i);
            __syncthreads();
        }
    };
    if (K > TOPK_CAP) {          // uniform
        materialise();
        topk_rounds(w, I, 0, K, oi, ov, rv, ri);
        return;
    }
    // ---- pre-filter.  The K-th largest of the 1024 per-thread maxima is a lower bound of the K-th largest score
    // (at least K scores reach it), and with scores dealt to threads in small runs only a few hundred do: the exact
    // selection then runs over those in LDS instead of histogramming every score four times.
    if constexpr (STREAM) if (streamed) {
        // Chunk c = items [c CH, (c+1) CH); thread t holds float4 number t + 1024 j of it, j < TOPK_RV4.
        constexpr int CH = 4096 * TOPK_RV4;
        const int nchunk = (I + CH - 1) / CH;
        float4 cur[TOPK_RV4], nxt[TOPK_RV4];
        auto load_chunk = [&](int c, float4* dst) {
#pragma unroll
            for (int j = 0; j < TOPK_RV4; ++j) {
                const int i4 = c * CH + 4 * (tid + 1024 * j);
                dst[j] = i4 < I ? *reinterpret_cast<const float4*>(s + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        load_chunk(0, cur);
        unsigned bound = 0u;
        bool overflow = false;
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk) load_chunk(c + 1, nxt);
            if (c == 0) {
                // the first chunk sets the bound: the (K + nseen)-th largest of the threads' RAW maxima -- at most nseen
                // of the maxima above it belong to seen items, so at least K unseen scores reach it
                unsigned m = 0u;
#pragma unroll
                for (int j = 0; j < TOPK_RV4; ++j) {
                    const int i4 = 4 * (tid + 1024 * j);
                    const float e[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned k = i4 + q < I ? topk_key(e[q]) : 0u;
                        m = k > m ? k : m;
                    }
                }
                tk[tid] = m;
                __syncthreads();
                topk_radix_select([&](int i) { return tk[i]; }, 1024, K + nseen, hist, &s_prefix, &s_remaining);
                bound = s_prefix;
            }
#pragma unroll
            for (int j = 0; j < TOPK_RV4; ++j) {
                const int i4 = c * CH + 4 * (tid + 1024 * j);
                const float e[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i4 + q;
                    if (i < I && topk_key(e[q]) >= bound) {
                        const float x = is_seen(i) ? e[q] - fabsf(e[q] * 1.0f) * 100000000.0f : e[q];
                        const unsigned k = topk_key(x);
                        if (k >= bound) {
                            const int pos = atomicAdd(&s_ncand, 1);
                            if (pos < TOPK_NCAND) { candk[pos] = k; candi[pos] = i; }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TOPK_RV4; ++j) cur[j] = nxt[j];
            if ((c & 3) == 3 && c + 1 < nchunk) {
                // every fourth chunk: does the list need tightening?  (uniform: read behind a barrier)
                __syncthreads();
                const int n = s_ncand;
                if (n > TOPK_NCAND) { overflow = true; break; }
                if (n > TOPK_NCAND / 2) {
                    topk_radix_select([&](int i) { return candk[i]; }, n, K, hist, &s_prefix, &s_remaining);
                    const unsigned T = s_prefix;
                    unsigned kk[4];
                    int ii[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = tid + 1024 * j;
                        kk[j] = i < n ? candk[i] : 0u;
                        ii[j] = i < n ? candi[i] : 0;
                    }
                    __syncthreads();
                    if (tid == 0) s_ncand = 0;
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (tid + 1024 * j < n && kk[j] >= T) {
                            const int pos = atomicAdd(&s_ncand, 1);
                            candk[pos] = kk[j];
                            candi[pos] = ii[j];
                        }
                    __syncthreads();
                    bound = T;
                    if (s_ncand > TOPK_NCAND / 2) { overflow = true; break; }     // (a flood of ties at the bound)
                }
            }
        }
        __syncthreads();
        if (!overflow && s_ncand > TOPK_NCAND) overflow = true;
        if (overflow) {          // uniform: start again on the legacy path
            __syncthreads();
            streamed = false;
            mask_setup();
            if (tid == 0) s_ncand = 0;
            __syncthreads();
        }
    }
    unsigned keys[TOPK_RV4][4];
    int wave_r = 65, nw_bound = 0;
    if (in_regs) {
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < TOPK_RV4; ++j) {
            const int i4 = 4 * (tid + 1024 * j);
            const float4 v = rowv[j];
            const float e[4] = {v.x, v.y, v.z, v.w};
            const unsigned mw = i4 < I ? maskw[i4 >> 5] >> (i4 & 31) : 0u;   // 4 | i4: the four bits share a word
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = ((mw >> c) & 1u) ? e[c] - fabsf(e[c] * 1.0f) * 100000000.0f : e[c];
                const unsigned k = i4 + c < I ? topk_key(x) : 0u;           // beyond the row: below every key
                keys[j][c] = k;
                m = k > m ? k : m;
            }
        }
        // The bound without a selection: the waves whose 64 lanes all hold items (the first nw) each find their r-th
        // largest maximum, r = ceil(K / nw), to 16 bits -- a binary search on the key's leading bits with ballot + count,
        // no LDS, no barrier: t ends as the r-th largest with its low half cleared, at least r lanes reach it -- and the
        // smallest of those is reached by at least nw r >= K maxima.  Looser than the K-th largest of all 1024 maxima
        // (about twice the candidates), and one barrier instead of the selection's eight.
        const int nw = I >= 253 ? min(16, (I - 253) / 256 + 1) : 0;
        wave_r = nw > 0 ? (K + nw - 1) / nw : 65;
        if (wave_r <= 64) {      // uniform
            unsigned kth = 0u;
#pragma unroll
            for (int b = 31; b >= 16; --b) {
                const unsigned t = kth | (1u << b);
                if (__popcll(__ballot(m >= t)) >= wave_r) kth = t;
            }
            if ((tid & 63) == 0) hist[tid >> 6] = (int)kth;
            nw_bound = nw;
        } else {
            tk[tid] = m;
        }
    } else if (!streamed) {
        unsigned m = 0u;
        for (int i = tid; i < I; i += 1024) { const unsigned k = topk_key(val(i)); m = k > m ? k : m; }
        tk[tid] = m;
    }
    if (!streamed) {
        __syncthreads();
        unsigned bound;
        if (nw_bound > 0) {
            bound = (unsigned)hist[0];
            for (int w = 1; w < nw_bound; ++w) bound = min(bound, (unsigned)hist[w]);
        } else {
            // (register-resident row: the bound from the two leading bytes of the maxima -- scores within 2^-7 relative
            //  below the K-th largest maximum join the candidates)
            topk_radix_select([&](int i) { return tk[i]; }, 1024, K, hist, &s_prefix, &s_remaining, in_regs ? 2 : 0);
            bound = s_prefix;
        }
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < TOPK_RV4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = 4 * (tid + 1024 * j) + c;
                    if (i < I && keys[j][c] >= bound) {
                        const int pos = atomicAdd(&s_ncand, 1);
                        if (pos < TOPK_NCAND) { candk[pos] = keys[j][c]; candi[pos] = i; }
                    }
                }
        } else {
            for (int i = tid; i < I; i += 1024) {
                const unsigned k = topk_key(val(i));
                if (k >= bound) {
                    const int pos = atomicAdd(&s_ncand, 1);
                    if (pos < TOPK_NCAND) { candk[pos] = k; candi[pos] = i; }
                }
            }
        }
        __syncthreads();
    }
    const int ncand = s_ncand;
    const bool filtered = ncand <= TOPK_NCAND;          // uniform; otherwise (heavy ties) select over all scores
    if (ncand <= 1024) {         // uniform -- the usual case (K = 100: about 200 candidates)
        // Rank by counting: a candidate's place in the stable descending order = how many candidates precede it (larger
        // key, or the same key and a lower item: ONE unsigned 64-bit comparison of (key, ~item)).  All 1024 threads
        // take part: thread = (candidate i, slice q of the others), slices by wave so that a wave's reads of the others
        // are broadcasts; the partial counts meet in LDS.  One barrier, no selection, no sort: this replaced the second
        // radix select (4 passes, 12 barriers), the collect and a 28-stage bitonic sort.
        int P2 = 64;
        while (P2 < ncand) P2 <<= 1;
        const int parts = 1024 / P2, i = tid & (P2 - 1), q = tid / P2;
        const int chunk = (((ncand + parts - 1) / parts) + 3) & ~3;
        const int j0 = q * chunk, j1 = min(ncand, j0 + chunk);
        if (i < ncand && j0 < j1) {
            const unsigned long long ca = ((unsigned long long)candk[i] << 32) | (unsigned)~candi[i];
            int rank = 0;
            int j = j0;
#define SV_PRECEDES(k_, i_) ((((unsigned long long)(k_) << 32) | (unsigned)~(i_)) > ca)
#pragma unroll 2
            for (; j + 4 <= j1; j += 4) {
                const uint4 kb = *reinterpret_cast<const uint4*>(&candk[j]);
                const int4 ib = *reinterpret_cast<const int4*>(&candi[j]);
                rank += SV_PRECEDES(kb.x, ib.x) + SV_PRECEDES(kb.y, ib.y) + SV_PRECEDES(kb.z, ib.z) + SV_PRECEDES(kb.w, ib.w);
            }
            for (; j < j1; ++j) rank += SV_PRECEDES(candk[j], candi[j]);
#undef SV_PRECEDES
            atomicAdd(&cidx[i], rank);
        }
        __syncthreads();
        if (tid < ncand) {
            const int rank = cidx[tid];
            if (rank < K) { oi[rank] = candi[tid]; ov[rank] = topk_unkey(candk[tid]); }
        }
        return;
    }
    if (filtered) topk_radix_select([&](int i) { return candk[i]; }, ncand, K, hist, &s_prefix, &s_remaining);
    else topk_radix_select([&](int i) { return topk_key(val(i)); }, I, K, hist, &s_prefix, &s_remaining);
    const unsigned T = s_prefix;
    // ---- collect: scores above T (fewer than K) and the items whose score equals T
    const int nsrc = filtered ? ncand : I;
    for (int i = tid; i < nsrc; i += 1024) {
        const unsigned k = filtered ? candk[i] : topk_key(val(i));
        const int it = filtered ? candi[i] : i;
        if (k > T) {
            const int pos = atomicAdd(&s_ngt, 1);
            ckey[pos] = k;
            cidx[pos] = it;
        } else if (k == T) {
            const int pos = atomicAdd(&s_nties, 1);
            if (pos < TOPK_CAP) tidx[pos] = it;
        }
    }
    __syncthreads();
    const int ngt = s_ngt, nties = s_nties;
    if (nties > TOPK_CAP) {      // uniform: too many ties at the threshold to sort here
        materialise();
        topk_rounds(w, I, 0, K, oi, ov, rv, ri);
        return;
    }
    // ---- bitonic sorts in LDS: candidates by (key descending, item ascending), ties by item ascending
    int n2 = 1;
    while (n2 < ngt) n2 <<= 1;
    for (int i = ngt + tid; i < n2; i += 1024) { ckey[i] = 0u; cidx[i] = 0x7fffffff; }      // pads sort last
    int m2 = 1;
    while (m2 < nties) m2 <<= 1;
    for (int i = nties + tid; i < m2; i += 1024) tidx[i] = 0x7fffffff;
    __syncthreads();
    if (n2 <= 128 && m2 <= 128) {
        // the usual case (K = 100): both lists fit one wave at two elements per lane, and a wave needs no workgroup
        // barrier between stages -- the LDS executes one wave's instructions in order (28 stages: 0.6 us against 7 us
        // of 16-wave barriers)
        if (tid < 64) {
            for (int size = 2; size <= n2 || size <= m2; size <<= 1) {
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const int i = ((tid & ~(stride - 1)) << 1) | (tid & (stride - 1)), j = i | stride;
                    const bool up = (i & size) == 0;
                    if (size <= n2 && j < n2) {
                        const unsigned ka = ckey[i], kb = ckey[j];
                        const int ia = cidx[i], ib = cidx[j];
                        const bool a_first = ka > kb || (ka == kb && ia < ib);
                        if (a_first != up) { ckey[i] = kb; ckey[j] = ka; cidx[i] = ib; cidx[j] = ia; }
                    }
                    if (size <= m2 && j < m2) {
                        const int ia = tidx[i], ib = tidx[j];
                        if ((ia < ib) != up) { tidx[i] = ib; tidx[j] = ia; }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __syncthreads();
    } else
    for (int size = 2; size <= n2 || size <= m2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int i = tid, j = i ^ stride;
            if (j > i) {
                const bool up = (i & size) == 0;         // this pair sorts "first before second"
                if (size <= n2 && j < n2) {
                    const unsigned ka = ckey[i], kb = ckey[j];
                    const int ia = cidx[i], ib = cidx[j];
                    const bool a_first = ka > kb || (ka == kb && ia < ib);
                    if (a_first != up) { ckey[i] = kb; ckey[j] = ka; cidx[i] = ib; cidx[j] = ia; }
                }
                if (size <= m2 && j < m2) {
                    const int ia = tidx[i], ib = tidx[j];
                    if ((ia < ib) != up) { tidx[i] = ib; tidx[j] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (int k = tid; k < K; k += 1024) {
        const int it = k < ngt ? cidx[k] : tidx[k - ngt];
        oi[k] = it;
        ov[k] = val(it);
    }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller N(0,1), keyed by (seed) and counter (row_key[r], col/4, step):
// replayable and independent of how users are sharded over GPUs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__global__ void philox_normal_kernel(float* __restrict__ out, const long long* __restrict__ row_key, int R,
                                     unsigned long long seed, unsigned step) {
    // thread -> (row, group of 4 columns); 16 groups per row
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)R * 16;
    if (i >= n) return;
    const int row = (int)(i >> 4), grp = (int)(i & 15);
    const unsigned long long key = (unsigned long long)row_key[row];
    uint32_t c[4] = {(uint32_t)key, (uint32_t)(key >> 32), (uint32_t)grp, step};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float4 o;
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    o.x = r0 * c0; o.y = r0 * s0; o.z = r1 * c1; o.w = r1 * s1;
    reinterpret_cast<float4*>(out)[i] = o;
}

inline int wave_rows_grid(int rows) {
    int b = (rows + 3) / 4;
    return b < 1 ? 1 : (b > 4096 ? 4096 : b);
}

}  // namespace

// Reference interface replaced: self.item_embed(x) (nb:756).
extern "C" int svae_embed_fwd(const float* table, const int* x, float* out, int R, int I, void* stream) {
    SV_CHECK_ARG(table && x && out && R > 0 && I > 0, "svae_embed_fwd: bad arguments");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(wave_rows_grid(R)), dim3(256), 0, (hipStream_t)stream, table, x, out, R, I);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: embedding backward inside loss.backward() (nb:1205).  gtable must be
// zero-filled by the caller; only rows of items present in the batch are written.  Segments are sorted by
// length (descending); the first nlong of them are the ones longer than svae_embed_long_rows().
extern "C" int svae_embed_long_rows(void) { return EMB_LONG; }
extern "C" int svae_embed_bwd(const float* demb, const int* perm, const int* seg_off, const int* seg_item,
                              float* gtable, int nseg, int nlong, float scale, void* stream) {
    SV_CHECK_ARG(demb && perm && seg_off && seg_item && gtable && nseg > 0 && nlong >= 0 && nlong <= nseg,
                 "svae_embed_bwd: bad arguments");
    const int nshort_wg = nseg > nlong ? wave_rows_grid(nseg - nlong) : 0;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(nlong + nshort_wg), dim3(256), 0, (hipStream_t)stream, demb, perm, seg_off,
                       seg_item, gtable, nlong, nseg, scale);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: Model.sample_latent (nb:734-750) + KL term of VAELoss (nb:797).
extern "C" int svae_latent_fwd(const float* tmp, const float* eps, float* z, float* kl_row, int R, void* stream) {
    SV_CHECK_ARG(tmp && eps && z && kl_row && R > 0, "svae_latent_fwd: bad arguments");
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(wave_rows_grid(R)), dim3(256), 0, (hipStream_t)stream, tmp, eps, z,
                       kl_row, R);
    SV_LAUNCH_CHECK();
    return 0;
}

extern "C" int svae_latent_bwd(const float* tmp, const float* eps, const float* dz, const float* w_kl, float anneal,
                               float* dtmp, int R, void* stream) {
    SV_CHECK_ARG(tmp && eps && dz && w_kl && dtmp && R > 0, "svae_latent_bwd: bad arguments");
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(wave_rows_grid(R)), dim3(256), 0, (hipStream_t)stream, tmp, eps, dz,
                       w_kl, anneal, dtmp, R);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: F.log_softmax + multinomial likelihood of VAELoss.forward (nb:802-808)
// and its gradient.  write_grad != 0 overwrites logits with dlogits.
extern "C" int svae_softmax_nll(float* logits, const int* tgt, const float* w_nll, float* nll_row, float* lse,
                                int R, int I, int ldI, int K, int write_grad, void* stream) {
    SV_CHECK_ARG(logits && tgt && nll_row && R > 0 && I > 0, "svae_softmax_nll: bad arguments");
    SV_CHECK_ARG(ldI >= I && (ldI % 4) == 0 && K >= 1, "svae_softmax_nll: bad ldI=%d / K=%d", ldI, K);
    SV_CHECK_ARG(!write_grad || w_nll, "svae_softmax_nll: w_nll required for the gradient");
    if (write_grad)
        hipLaunchKernelGGL(softmax_nll_kernel<true>, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, tgt, w_nll,
                           nll_row, lse, I, ldI, K);
    else
        hipLaunchKernelGGL(softmax_nll_kernel<false>, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, tgt, w_nll,
                           nll_row, lse, I, ldI, K);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: optimizer.step() with torch.optim.Adam(weight_decay=5e-3) (nb:1206, nb:1260-1263).
// step is the 1-based step count; gscale multiplies the raw gradient first (1/world for data parallel sums).
extern "C" int svae_adam_step(float* p, const float* g, float* m, float* v, long long n, int step, float lr, float b1,
                              float b2, float eps, float wd, float gscale, void* stream) {
    SV_CHECK_ARG(p && g && m && v && n > 0 && (n % 4) == 0 && step >= 1, "svae_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, b1, b2, step_size,
                       inv_sqrt_bc2, eps, wd, gscale);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: optimizer.step() with the notebook's other optimizer choices (nb:1252-1259,
// nb:1264-1267).  kind 0 = torch.optim.Adagrad (a = state_sum; decay = lr_decay; step 1-based), 1 = Adadelta
// (a = square_avg, b = acc_delta; decay = rho), 2 = RMSprop (a = square_avg; decay = alpha; plain: no momentum,
// not centered).  b may be NULL unless kind == 1.
extern "C" int svae_optimizer_step(int kind, float* p, const float* g, float* a, float* b, long long n, int step,
                                   float lr, double decay, float eps, float wd, float gscale, void* stream) {
    SV_CHECK_ARG(kind >= 0 && kind <= 2, "svae_optimizer_step: kind must be 0 (adagrad), 1 (adadelta) or 2 (rmsprop)");
    SV_CHECK_ARG(p && g && a && (kind != 1 || b) && n > 0 && (n % 4) == 0 && step >= 1,
                 "svae_optimizer_step: bad arguments");
    long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    const float d = (float)decay, omd = (float)(1.0 - decay);
    if (kind == 0) {
        const float clr = (float)((double)lr / (1.0 + (double)(step - 1) * decay));
        hipLaunchKernelGGL(opt_kernel<0>, dim3(blocks), dim3(256), 0, st, p, g, a, b, n4, clr, 0.0f, 0.0f, eps, wd,
                           gscale);
    } else if (kind == 1) {
        hipLaunchKernelGGL(opt_kernel<1>, dim3(blocks), dim3(256), 0, st, p, g, a, b, n4, lr, d, omd, eps, wd, gscale);
    } else {
        hipLaunchKernelGGL(opt_kernel<2>, dim3(blocks), dim3(256), 0, st, p, g, a, b, n4, lr, d, omd, eps, wd, gscale);
    }
    SV_LAUNCH_CHECK();
    return 0;
}

extern "C" int svae_colsum(const float* a, float* out, float* partials, int R, int N, int ld, int nparts,
                           void* stream) {
    SV_CHECK_ARG(a && out && partials && R > 0 && N > 0 && ld >= N && nparts >= 1, "svae_colsum: bad arguments");
    const int rpb = (R + nparts - 1) / nparts;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, nparts), dim3(256), 0, (hipStream_t)stream, a,
                       partials, R, N, ld, rpb);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, partials, out, N,
                       nparts);
    SV_LAUNCH_CHECK();
    return 0;
}

extern "C" int svae_transpose(const float* in, float* out, int rows, int cols, void* stream) {
    SV_CHECK_ARG(in && out && rows > 0 && cols > 0, "svae_transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream,
                       in, out, rows, cols);
    SV_LAUNCH_CHECK();
    return 0;
}

// out3 = {total loss, likelihood part, un-annealed KL part}
extern "C" int svae_loss_reduce(const float* nll_row, const float* kl_row, const float* w_nll, const float* w_kl,
                                float anneal, int R, float* out3, void* stream) {
    SV_CHECK_ARG(nll_row && kl_row && w_nll && w_kl && out3 && R > 0, "svae_loss_reduce: bad arguments");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nll_row, kl_row, w_nll, w_kl,
                       anneal, R, out3);
    SV_LAUNCH_CHECK();
    return 0;
}

extern "C" int svae_gather_rows(const float* in, const int* idx, float* out, int nrows, int ld_in, int ld_out, int n,
                                void* stream) {
    SV_CHECK_ARG(in && idx && out && nrows > 0 && (ld_in % 4) == 0 && (ld_out % 4) == 0 && (n % 4) == 0,
                 "svae_gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, in, idx, out, nrows,
                       ld_in / 4, ld_out / 4, n / 4);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: seen-item mask + torch.sort(-scores)[:k] in evaluate() (nb:621-636).
static void topk_launch(const float* scores, int ld, const int* seen, const int* seen_off, int n_users, int I, int K,
                        int* out_idx, float* out_val, float* work, hipStream_t st) {
    if (I > 4096 * TOPK_RV4)
        hipLaunchKernelGGL(masked_topk_kernel<true>, dim3(n_users), dim3(1024), 0, st, scores, ld, seen, seen_off, I, K,
                           out_idx, out_val, work);
    else
        hipLaunchKernelGGL(masked_topk_kernel<false>, dim3(n_users), dim3(1024), 0, st, scores, ld, seen, seen_off, I, K,
                           out_idx, out_val, work);
}

// work is a scratch copy of the scores, [n_users][ld].
extern "C" int svae_masked_topk(const float* scores, int ld, const int* seen, const int* seen_off, int n_users, int I,
                                int K, int* out_idx, float* out_val, float* work, void* stream) {
    SV_CHECK_ARG(scores && seen && seen_off && out_idx && out_val && work, "svae_masked_topk: null pointer");
    SV_CHECK_ARG(n_users > 0 && I > 0 && ld >= I && K >= 1 && K <= I, "svae_masked_topk: bad shape");
    topk_launch(scores, ld, seen, seen_off, n_users, I, K, out_idx, out_val, work, (hipStream_t)stream);
    SV_LAUNCH_CHECK();
    return 0;
}

// ---- evaluate()'s membership search ---------------------------------------------------------------------------
// Reference interface replaced: the hit search of evaluate(), nb:636-645 (`rec_list = argsorted[:k].cpu()`, then
// `if movie in rec_list` / `rec_list.index(movie)` for every held-out item: an O(n_heldout * k) Python list search per
// user behind a device-to-host copy).  Here: one workgroup per scored user; the user's top-k ids go to LDS, every held-out
// item of the user gets its 0-based position in that list, or -1.  The user's GLOBAL index (into held_off) is taken
// from the noise key of its last row (row_key = user_key << 20 | timestep, svae_plan_create): plans sort their users by
// length, and the key is what says who is who.  Integer work, bit-exact; the float64 metric arithmetic of nb:637-654
// stays on the host (one copy of rank_out for the whole evaluation instead of one [users, k] copy per batch).
__global__ __launch_bounds__(64) void heldout_rank_kernel(const int* __restrict__ idx, int k,
                                                           const long long* __restrict__ row_key,
                                                           const int* __restrict__ last_row,
                                                           const int* __restrict__ held,
                                                           const long long* __restrict__ held_off, long long n_total_users,
                                                           int* __restrict__ rank_out) {
    extern __shared__ int top[];
    const int u = blockIdx.x;
    for (int i = threadIdx.x; i < k; i += 64) top[i] = idx[(size_t)u * k + i];
    __syncthreads();
    const long long g = row_key[last_row[u]] >> 20;
    if (g < 0 || g >= n_total_users) return;        // (not a user of this evaluation: the host checks its keys)
    const long long a = held_off[g], b = held_off[g + 1];
    for (long long j = a + threadIdx.x; j < b; j += 64) {
        const int h = held[j];
        int r = -1;
        for (int i = 0; i < k; ++i)
            if (top[i] == h) { r = i; break; }
        rank_out[j] = r;
    }
}
extern "C" int svae_heldout_ranks(const int* topk_idx, int n_users, int K, const long long* row_key, const int* last_row,
                                  const int* held, const long long* held_off, long long n_total_users, int* rank_out,
                                  void* stream) {
    SV_CHECK_ARG(topk_idx && row_key && last_row && held && held_off && rank_out, "svae_heldout_ranks: null pointer");
    SV_CHECK_ARG(n_users > 0 && K >= 1 && K <= 4096 && n_total_users >= n_users, "svae_heldout_ranks: bad shape");
    hipLaunchKernelGGL(heldout_rank_kernel, dim3(n_users), dim3(64), K * sizeof(int), (hipStream_t)stream, topk_idx, K,
                       row_key, last_row, held, held_off, n_total_users, rank_out);
    SV_LAUNCH_CHECK();
    return 0;
}

// ---- sparse exchange of the embedding gradient: the rebuild behind the all-gather -------------------------------
// No counterpart in the reference (single process).  Data parallel runs with large catalogues exchange the item-embedding
// gradient as (row id, row) pairs (dist.GradSync.enable_sparse_rows): after the all-gather every rank holds world lists
// of cap pairs (id -1 = padding) and rebuilds the SUMMED rows in its own gradient table: its own rows zeroed first, then
// rank 0's rows added, then rank 1's, ... -- one launch per rank on one stream, so the additions happen in the same
// order on every rank (replicas stay bit-identical) and, ids being distinct within a rank's list, without atomics.
// One wave per pair; rows of `width` floats (a multiple of 4).  Replaces torch's index_fill_ / index_add_ on the product
// path (round 2's review), pad entries are skipped instead of being multiplied by zero.
__global__ __launch_bounds__(256) void sparse_rows_zero_kernel(float* __restrict__ table, int width,
                                                              const long long* __restrict__ ids, int n, long long rows) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const long long id = ids[i];
    if (id < 0 || id >= rows) return;
    float4* dst = reinterpret_cast<float4*>(table + (size_t)id * width);
    for (int c = lane; c < width / 4; c += 64) dst[c] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void sparse_rows_add_kernel(float* __restrict__ table, int width,
                                                             const long long* __restrict__ ids,
                                                             const float* __restrict__ vals, int n, long long rows) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const long long id = ids[i];
    if (id < 0 || id >= rows) return;
    float4* dst = reinterpret_cast<float4*>(table + (size_t)id * width);
    const float4* src = reinterpret_cast<const float4*>(vals + (size_t)i * width);
    for (int c = lane; c < width / 4; c += 64) {
        float4 a = dst[c];
        const float4 b = src[c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[c] = a;
    }
}
extern "C" int svae_sparse_rows_rebuild(float* table, long long rows, int width, const long long* all_ids,
                                        const float* all_vals, int world, int cap, const long long* own_ids, int n_own,
                                        void* stream) {
    SV_CHECK_ARG(table && all_ids && all_vals && (own_ids || n_own == 0), "svae_sparse_rows_rebuild: null pointer");
    SV_CHECK_ARG(rows > 0 && width > 0 && (width % 4) == 0 && world >= 1 && cap >= 1 && n_own >= 0 && n_own <= cap,
                 "svae_sparse_rows_rebuild: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (n_own > 0)
        hipLaunchKernelGGL(sparse_rows_zero_kernel, dim3((n_own + 3) / 4), dim3(256), 0, st, table, width, own_ids, n_own, rows);
    for (int r = 0; r < world; ++r)
        hipLaunchKernelGGL(sparse_rows_add_kernel, dim3((cap + 3) / 4), dim3(256), 0, st, table, width,
                           all_ids + (size_t)r * cap, all_vals + (size_t)r * cap * width, cap, rows);
    SV_LAUNCH_CHECK();
    return 0;
}

// ---- the same top-k for callers that cannot allocate (no counterpart in the reference) -----------------------
// svae_masked_topk takes its scratch copy of the scores (`work`) from the caller, who has to allocate it.  Code under
// HIP graph capture, a tight serving loop or a harness that forbids hipMalloc cannot: svae_masked_topk_static runs the
// same kernel with the scratch in a file-scope device array.  Every GPU the library runs on holds its own copy of the
// array (32 MiB) at its own address until the process ends; the address is looked up once per device, with that device
// current (svae_scratch_setup, called by svae_check_device and, if that was skipped, by the first call on a device).
// In host code the array's NAME is only a handle for hipGetSymbolAddress: kernels get the looked-up pointer.
constexpr long long TOPK_STATIC_FLOATS = 8ll << 20;
static __device__ __align__(256) float scratch[TOPK_STATIC_FLOATS];

constexpr int SCRATCH_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static float* g_scratch_ptr[SCRATCH_MAX_DEVICES] = {nullptr};

// Address of `scratch` on the CURRENT device (looked up on first use; no allocation: the array is part of the loaded
// code object).  Returns 0 and sets *out, or a svae_fail code.
int svae_scratch_setup(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return svae_fail((int)e, "svae_scratch_setup: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return svae_fail(-1, "svae_scratch_setup: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr)
            return svae_fail(e != hipSuccess ? (int)e : -3, "svae_scratch_setup: hipGetSymbolAddress(scratch) on device %d: %s",
                             dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    if (out) *out = g_scratch_ptr[dev];
    return 0;
}

// No counterpart in the reference: svae_masked_topk (nb:621-636) for callers that cannot allocate.  Same arguments
// without `work`; n_users * ld must not exceed the static scratch (8 Mi floats).  The stream must belong to the current
// device.  Calls that overlap on one GPU (two streams) race on the array; nothing in it has to start at zero (the
// kernel writes every element of a row's copy before it reads it), so the last call's contents do not matter.
extern "C" int svae_masked_topk_static(const float* scores, int ld, const int* seen, const int* seen_off, int n_users,
                                       int I, int K, int* out_idx, float* out_val, void* stream) {
    SV_CHECK_ARG(scores && seen && seen_off && out_idx && out_val, "svae_masked_topk_static: null pointer");
    SV_CHECK_ARG(n_users > 0 && I > 0 && ld >= I && K >= 1 && K <= I, "svae_masked_topk_static: bad shape");
    SV_CHECK_ARG((long long)n_users * (long long)ld <= TOPK_STATIC_FLOATS,
                 "svae_masked_topk_static: n_users * ld = %lld floats exceeds the static scratch of %lld floats "
                 "(32 MiB): score fewer users per call or use svae_masked_topk with a caller-owned work buffer",
                 (long long)n_users * (long long)ld, TOPK_STATIC_FLOATS);
    float* work = nullptr;
    const int rc = svae_scratch_setup(&work);
    if (rc != 0) return rc;
    topk_launch(scores, ld, seen, seen_off, n_users, I, K, out_idx, out_val, work, (hipStream_t)stream);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: np.random.normal(0, 1, size) in sample_latent (nb:744), as a counter-based stream.
extern "C" int svae_philox_normal(float* out, const long long* row_key, int R, unsigned long long seed,
                                  unsigned step, void* stream) {
    SV_CHECK_ARG(out && row_key && R > 0, "svae_philox_normal: bad arguments");
    const long long n = (long long)R * 16;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                       row_key, R, seed, step);
    SV_LAUNCH_CHECK();
    return 0;
}

// Stream calibration helper (no counterpart in the reference): one workgroup that stays busy for `usec`
// microseconds of the 100 MHz wall clock.  The host launches it on two streams to find out whether they run
// concurrently: HIP maps streams onto a few hardware queues, and two streams that share a queue execute in
// order, whatever the stream graph says (engine.py, _pick_side_stream).  Bounded loop: the kernel always ends.
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 22); ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
extern "C" int svae_spin(int usec, void* stream) {
    SV_CHECK_ARG(usec > 0 && usec <= 100000, "svae_spin: usec out of range");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull);
    SV_LAUNCH_CHECK();
    return 0;
}

// A grid of `nwg` one-wave workgroups, each busy for `usec` microseconds and holding `lds_bytes` of LDS (so that only a
// few fit a CU: the launch keeps its queue's dispatcher busy for many rounds while leaving most of the chip free).
__global__ void spin_grid_kernel(unsigned long long ticks) {
    extern __shared__ unsigned char spin_lds[];
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 20); ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (ticks == ~0ull) spin_lds[threadIdx.x] = 1;      // (keeps the allocation)
}
extern "C" int svae_spin_grid(int usec, int nwg, int lds_bytes, void* stream) {
    SV_CHECK_ARG(usec > 0 && usec <= 10000 && nwg > 0 && nwg <= (1 << 20) && lds_bytes >= 0 && lds_bytes <= 65536,
                 "svae_spin_grid: arguments out of range");
    hipLaunchKernelGGL(spin_grid_kernel, dim3(nwg), dim3(64), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)usec * 100ull);
    SV_LAUNCH_CHECK();
    return 0;
}

// One thread writes the device wall clock (100 MHz) into buf[slot]: in-stream time stamps for tools/step_stamps.py (where
// does the caller's stream wait inside a step -- without a profiler attached, whose launch cost distorts exactly that).
__global__ void stamp_kernel(unsigned long long* buf, int slot) { buf[slot] = wall_clock64(); }
extern "C" int svae_stamp(unsigned long long* buf, int slot, void* stream) {
    SV_CHECK_ARG(buf && slot >= 0, "svae_stamp: bad arguments");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, slot);
    SV_LAUNCH_CHECK();
    return 0;
}

// Zero fill (replaces the zero_grad pair nb:1197-1198 for the one gradient that is accumulated into, the embedding table's).
extern "C" int svae_zero(void* p, long long nbytes, void* stream) {
    SV_CHECK_ARG(p && nbytes > 0, "svae_zero: bad arguments");
    hipError_t e = hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream);
    if (e != hipSuccess) return svae_fail((int)e, "svae_zero: %s", hipGetErrorString(e));
    return 0;
}

// Backward / optimizer kernels of the training step (reference: torch autograd + torch.optim.AdamW(fused) over
// the same modules; train.py:150-213, utils/training_utils.py:38-52).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float ld1(const T* p) { return Elem<T>::load(p); }

// sum over the 8 lanes that share lane >> 3 (one (token, head) row of 64 values, 8 per lane)
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// ----------------------------------------------------------------------------------- cast
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, long ld_in, TO* __restrict__ out, long ld_out,
                                                   int rows, int cols) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cols, c = i % cols;
        Elem<TO>::store(out + r * ld_out + c, Elem<TI>::load(in + r * ld_in + c));
    }
}
// cols, both leading dimensions and both base addresses multiples of 8 elements / 16 bytes: 8 values per thread
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast8_kernel(const TI* __restrict__ in, long ld_in, TO* __restrict__ out, long ld_out,
                                                    int rows, int cols8) {
    const long n = (long)rows * cols8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cols8, c = (i - r * cols8) * 8;
        float v[8];
        load8(in + r * ld_in + c, v);
        store8(out + r * ld_out + c, v);
    }
}

// ----------------------------------------------------------------------------------- attention backward
// D[b,h,q] = sum_d dO[q, h*64+d] * O[q, h*64+d]   (token-major O / dO; 8 lanes x 8 values per (token, head), 8 of them per wave)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ O, const T* __restrict__ dO, long ld,
                                                         float* __restrict__ D, int B, int H, int L) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const long w = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool ok = w < (long)B * L * H;
    const long wc = ok ? w : 0;
    const int h = (int)(wc % H);
    const long row = wc / H;                      // b * L + l
    float a[8], b[8];
    load8(O + row * ld + h * 64 + sub * 8, a);
    load8(dO + row * ld + h * 64 + sub * 8, b);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v += a[i] * b[i];
    const float s = group8_sum(v);
    if (ok && sub == 0) D[((row / L) * H + h) * L + row % L] = s;
}

// Reference-quality backward (fp32 arithmetic, one wave per row, lane = head dimension).  Used by the fp32 parity
// mode and as the on-device check of the MFMA kernels.  Qs carries scale*log2(e) (scores are log2-domain):
//   P = exp2(Qs.K - lse), dS = P * (dO.V - D);  dqhat = scale * dS K  (gradient w.r.t. the UNSCALED normalised q),
//   dK = ln2 * dS^T Qs, dV = P^T dO.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ Qs, long q_bstride, const T* __restrict__ K,
                                                          const T* __restrict__ V, const T* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ D,
                                                          T* __restrict__ dQ, int H, int Lq, int Lk, float scale) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Lq) return;
    const long bh = (long)b * H + h;
    const float qv = ld1(Qs + (long)b * q_bstride + ((long)h * Lq + q) * 64 + lane);
    const float dov = ld1(dO + (bh * Lq + q) * 64 + lane);
    const float l2 = lse[bh * Lq + q], dl = D[bh * Lq + q];
    const T* Kh = K + bh * (long)Lk * 64;
    const T* Vh = V + bh * (long)Lk * 64;
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) {
        const float kv = ld1(Kh + (long)j * 64 + lane), vv = ld1(Vh + (long)j * 64 + lane);
        const float s2 = wave_sum(qv * kv);
        const float dp = wave_sum(dov * vv);
        const float ds = exp2f(s2 - l2) * (dp - dl);
        acc += ds * kv;
    }
    Elem<T>::store(dQ + (bh * Lq + q) * 64 + lane, acc * scale);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ Qs, long q_bstride, const T* __restrict__ K,
                                                           const T* __restrict__ V, const T* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ D,
                                                           T* __restrict__ dK, T* __restrict__ dV, int H, int Lq, int Lk) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.z, h = blockIdx.y;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= Lk) return;
    const long bh = (long)b * H + h;
    const float kv = ld1(K + (bh * Lk + j) * 64 + lane), vv = ld1(V + (bh * Lk + j) * 64 + lane);
    const T* Qh = Qs + (long)b * q_bstride + (long)h * Lq * 64;
    const T* dOh = dO + bh * (long)Lq * 64;
    float dk = 0.f, dv = 0.f;
    for (int i = 0; i < Lq; ++i) {
        const float qv = ld1(Qh + (long)i * 64 + lane), dov = ld1(dOh + (long)i * 64 + lane);
        const float p = exp2f(wave_sum(qv * kv) - lse[bh * Lq + i]);
        const float ds = p * (wave_sum(dov * vv) - D[bh * Lq + i]);
        dv += p * dov;
        dk += ds * qv;
    }
    Elem<T>::store(dK + (bh * Lk + j) * 64 + lane, dk * 0.69314718055994530942f);
    Elem<T>::store(dV + (bh * Lk + j) * 64 + lane, dv);
}

// ----------------------------------------------------------------------------------- qkv split backward
// Head-major gradients (dQhat, dKhat: w.r.t. the RMS-normalised q / k; dV) -> token-major d(qkv); RMSNorm backward for
// q and k needs the raw projections.  A (token, head) row of 64 values = 8 lanes x 8 values (16-byte accesses, 128-byte lines), 8 rows
// per wave; the per-row sums are three xor-shuffles.  w-gradients: per-workgroup partial sums [q 64 | k 64].
//   y = x * r * w, r = rsqrt(mean(x^2) + eps):  dx = r * (g - xh * mean(g * xh)), g = dy * w, xh = x * r ; dw += dy * xh
template <typename T>
__global__ __launch_bounds__(256) void qkv_split_bwd_kernel(const T* __restrict__ dQ, const T* __restrict__ dK, const T* __restrict__ dV,
                                                            const T* __restrict__ q_raw, long ldq, const T* __restrict__ k_raw, long ldk,
                                                            const float* __restrict__ qw, const float* __restrict__ kw, float eps,
                                                            T* __restrict__ dq_out, long ldoq, T* __restrict__ dk_out, long ldok,
                                                            T* __restrict__ dv_out, long ldov, float* __restrict__ partial, int B, int L,
                                                            int H) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane & 7, c0 = sub * 8;
    float pq[8], pk[8], wq[8], wk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pq[i] = 0.f; pk[i] = 0.f;
        wq[i] = qw ? qw[c0 + i] : 1.f;
        wk[i] = kw ? kw[c0 + i] : 1.f;
    }
    const long total = (long)B * L * H;
    for (long w0 = ((long)blockIdx.x * 4 + wv) * 8; w0 < total; w0 += (long)gridDim.x * 32) {
        const long w = w0 + (lane >> 3);
        const bool ok = w < total;
        const long wc = ok ? w : total - 1;       // clamped lanes compute on a valid row and store nothing
        const int h = (int)(wc % H);
        const long row = wc / H;                  // b * L + l
        const long b = row / L, l = row % L;
        const long hm = ((b * H + h) * L + l) * 64 + c0;
        auto norm_bwd = [&](const T* dy_p, const T* raw, long ld, const float* wgt, const float (&wv8)[8], T* out, long ldo,
                            float (&pw)[8]) {
            float dy[8];
            load8(dy_p + hm, dy);
            T* o = out + row * ldo + h * 64 + c0;
            if (!wgt) { if (ok) store8(o, dy); return; }
            float x[8];
            load8(raw + row * ld + h * 64 + c0, x);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
            const float r = rsqrtf(group8_sum(ss) * (1.0f / 64.0f) + eps);
            float g[8], mm = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] *= r;                         // xh
                g[i] = dy[i] * wv8[i];
                mm += g[i] * x[i];
            }
            const float m = group8_sum(mm) * (1.0f / 64.0f);
            float res[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (ok) pw[i] += dy[i] * x[i];
                res[i] = r * (g[i] - x[i] * m);
            }
            if (ok) store8(o, res);
        };
        if (dQ) norm_bwd(dQ, q_raw, ldq, qw, wq, dq_out, ldoq, pq);
        if (dK) norm_bwd(dK, k_raw, ldk, kw, wk, dk_out, ldok, pk);
        if (dV) {
            float v[8];
            load8(dV + hm, v);
            if (ok) store8(dv_out + row * ldov + h * 64 + c0, v);
        }
    }
    // the 8 rows of a wave (lanes with equal sub), then the 4 waves, in a fixed order
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = pq[i], c = pk[i];
        a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        c += __shfl_xor(c, 8, 64); c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
        if (lane < 8) { red[0][wv][c0 + i] = a; red[1][wv][c0 + i] = c; }
    }
    __syncthreads();
    if (wv == 0) {
        partial[(long)blockIdx.x * 128 + lane] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        partial[(long)blockIdx.x * 128 + 64 + lane] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    }
}

// ----------------------------------------------------------------------------------- 3-wide head backward
// out = A W^T + b, W [3,K]:  dA[m,k] = sum_j dout[m,j] W[j,k] ; partial[blk][3*K] = sum_m dout[m,j] A[m,k]
template <typename T>
__global__ __launch_bounds__(256) void linear_n3_bwd_kernel(const T* __restrict__ A, long lda, const float* __restrict__ W,
                                                            const float* __restrict__ dout, T* __restrict__ dA, long ldda,
                                                            float* __restrict__ partial, int M, int K, const T* __restrict__ mul, long ldmul) {
    const int t = threadIdx.x;
    // each thread owns columns t, t+256, ... ; rows are strided over blocks
    for (int k = t; k < K; k += 256) {
        const float w0 = W[k], w1 = W[K + k], w2 = W[2 * (long)K + k];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (long m = blockIdx.x; m < M; m += gridDim.x) {
            const float d0 = dout[m * 3], d1 = dout[m * 3 + 1], d2 = dout[m * 3 + 2];
            const float a = ld1(A + m * lda + k);
            s0 += d0 * a; s1 += d1 * a; s2 += d2 * a;
            const float f = mul ? ld1(mul + m * ldmul + k) : 1.0f;
            Elem<T>::store(dA + m * ldda + k, (d0 * w0 + d1 * w1 + d2 * w2) * f);
        }
        float* pr = partial + (long)blockIdx.x * 3 * K;
        pr[k] = s0; pr[K + k] = s1; pr[2 * (long)K + k] = s2;
    }
}

// The same with 16-byte accesses (K % 8 == 0, K <= 2048, aligned rows): thread t owns the 8-column chunk t % (K / 8) of the rows
// rs, rs + nrs, ... of its workgroup's share (nrs = 256 / (K / 8) rows side by side; K = 768: 96 chunks x 2 rows, 192 busy
// threads), four rows in flight per thread; the row slots' sums meet in LDS in a fixed order.  The scalar kernel above moved two
// bytes per lane and access: 134 us for the 49152 x 768 head of one training sample (150 MB), this one ~35.
template <typename T>
__global__ __launch_bounds__(256) void linear_n3_bwd_vec_kernel(const T* __restrict__ A, long lda, const float* __restrict__ W,
                                                                const float* __restrict__ dout, T* __restrict__ dA, long ldda,
                                                                float* __restrict__ partial, int M, int K, const T* __restrict__ mul, long ldmul) {
    extern __shared__ float red[];              // [nrs - 1][3][K]
    const int nch = K >> 3, nrs = 256 / nch, t = threadIdx.x;
    const int ch = t % nch, rs = t / nch;
    float w[3][8], sum[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        load8(W + (long)j * K + ch * 8, w[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[j][e] = 0.f;
    }
    if (rs < nrs) {
        const long step = (long)gridDim.x * nrs;
        auto row = [&](long m, const float (&a)[8]) {
            const float d0 = dout[m * 3], d1 = dout[m * 3 + 1], d2 = dout[m * 3 + 2];
            float o[8], f[8];
            if (mul) load8(mul + m * ldmul + ch * 8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sum[0][e] += d0 * a[e]; sum[1][e] += d1 * a[e]; sum[2][e] += d2 * a[e];
                o[e] = d0 * w[0][e] + d1 * w[1][e] + d2 * w[2][e];
                if (mul) o[e] *= f[e];
            }
            store8(dA + m * ldda + ch * 8, o);
        };
        long m = (long)blockIdx.x * nrs + rs;
        for (; m + 3 * step < M; m += 4 * step) {
            float a0[8], a1[8], a2[8], a3[8];
            load8(A + m * lda + ch * 8, a0); load8(A + (m + step) * lda + ch * 8, a1);
            load8(A + (m + 2 * step) * lda + ch * 8, a2); load8(A + (m + 3 * step) * lda + ch * 8, a3);
            row(m, a0); row(m + step, a1); row(m + 2 * step, a2); row(m + 3 * step, a3);
        }
        for (; m < M; m += step) {
            float a0[8];
            load8(A + m * lda + ch * 8, a0);
            row(m, a0);
        }
        if (rs > 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[((long)(rs - 1) * 3 + j) * K + ch * 8 + e] = sum[j][e];
        }
    }
    __syncthreads();
    if (rs == 0) {
        float* pr = partial + (long)blockIdx.x * 3 * K;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            for (int r = 1; r < nrs; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[j][e] += red[((long)(r - 1) * 3 + j) * K + ch * 8 + e];
            store8(pr + (long)j * K + ch * 8, sum[j]);
        }
    }
}

// ----------------------------------------------------------------------------------- loss backward
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ gscale, float coef, float* __restrict__ d, long n) {
    const float g = coef * (gscale ? *gscale : 1.0f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = g * (pred[i] - target[i]);
}

// ----------------------------------------------------------------------------------- optimizer
// torch.optim.AdamW semantics (decoupled weight decay): p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  `gscale` (device scalar, may be NULL) multiplies the
// gradient first (gradient clipping coefficient), `nan_to_num` applies train.py:181-183 (nan -> 0, +-inf -> +-1e-6).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi;
    }
}

// One launch over the optimizer's whole flat parameter buffer (fused multi-tensor AdamW): elements [0, n_decay) carry
// weight decay (the parameters with dim() > 1), the rest do not; n and n_decay are multiples of 4 (16-byte segments).
__global__ __launch_bounds__(256) void adamw_flat_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                         float4* __restrict__ v, long n4, long n4_decay, float lr, float b1, float b2,
                                                         float eps, float wd, float bc1, float bc2_sqrt,
                                                         const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.0f;
    const float step = lr / bc1, inv_bc2 = 1.0f / bc2_sqrt;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float keep = i < n4_decay ? 1.0f - lr * wd : 1.0f;
        const float4 gi = g[i], mo = m[i], vo = v[i];
        float4 pi = p[i], mi, vi;
#define M324_ADAM1(c)                                                     \
        {                                                                 \
            const float gc = gi.c * gs;                                   \
            mi.c = b1 * mo.c + (1.0f - b1) * gc;                          \
            vi.c = b2 * vo.c + (1.0f - b2) * gc * gc;                     \
            pi.c = pi.c * keep - step * mi.c / (sqrtf(vi.c) * inv_bc2 + eps); \
        }
        M324_ADAM1(x) M324_ADAM1(y) M324_ADAM1(z) M324_ADAM1(w)
#undef M324_ADAM1
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}

// bf16 copies of the optimizer's flat fp32 parameters, row-major AND transposed, in ONE launch (m324_weight_mirror): the forward /
// dgrad GEMM operands of every Linear weight after an update.  One workgroup per 64 x 64 tile; the item of a tile by binary search
// over the items' first-tile numbers (the table lives in device memory: it never changes between steps).
__global__ __launch_bounds__(256) void weight_mirror_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, bf16_t* __restrict__ dstT,
                                                            const m324_mirror_item* __restrict__ items, int n_items) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_items - 1;
    const long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_tile <= b) lo = mid; else hi = mid - 1;
    }
    const m324_mirror_item it = items[lo];
    const int tc = it.cols >> 6, tl = (int)(b - it.first_tile);          // cols is a multiple of 64: every column tile is whole
    const int r0 = (tl / tc) * 64, c0 = (tl % tc) * 64, t = threadIdx.x;
    const float* s = src + it.src_off;
    bf16_t* d = dst + it.dst_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {               // 16 rows per pass: a lane takes four consecutive values (16 bytes in, 8 out)
        const int r = 16 * i + (t >> 4), c = (t & 15) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < it.rows) {
            x = *reinterpret_cast<const float4*>(s + (long)(r0 + r) * it.cols + c0 + c);
            *reinterpret_cast<uint2*>(d + (long)(r0 + r) * it.cols + c0 + c) = make_uint2(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w));
        }
        tile[r][c] = x.x; tile[r][c + 1] = x.y; tile[r][c + 2] = x.z; tile[r][c + 3] = x.w;
    }
    if (it.dstT_off < 0) return;
    __syncthreads();
    bf16_t* dT = dstT + it.dstT_off;
#pragma unroll
    for (int i = 0; i < 8; ++i) {               // 8 transposed rows per pass: a lane packs two neighbouring source rows (4 bytes out)
        const int c = 8 * i + (t >> 5), r = (t & 31) * 2;
        if (r0 + r < it.rows)                   // (a row count is never odd inside a tile that has its second row: rows past the end are zeros
            *reinterpret_cast<uint32_t*>(dT + (long)(c0 + c) * it.ldT + r0 + r) = pack_bf16x2(tile[r][c], tile[r + 1][c]);      //  and land in the pad)
    }
}

__device__ __forceinline__ float sanitized(float x, bool& changed) {      // nan_to_num(0, 1e-6, -1e-6), train.py:181-183
    if (x != x) { changed = true; return 0.f; }
    if (x == INFINITY) { changed = true; return 1e-6f; }
    if (x == -INFINITY) { changed = true; return -1e-6f; }
    return x;
}

// VEC: 16-byte accesses, four of them in flight per lane and pass (the 628 MB flat gradient buffer in one launch of at most 1024
// workgroups: with one 4-byte load per lane and pass the kernel ran at 1 TB/s), and a value is written back only when the
// sanitizer changed it (never, in a healthy step).  Fixed summation order per lane, wave, workgroup: deterministic.
template <bool VEC>
__global__ __launch_bounds__(256) void grad_sanitize_sumsq_kernel(float* __restrict__ g, long n, int sanitize,
                                                                  float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    if (VEC) {
        float4* g4 = reinterpret_cast<float4*>(g);
        const long n4 = n >> 2;
        auto one = [&](long i, float4 x) {
            if (sanitize) {
                bool ch = false;
                x.x = sanitized(x.x, ch); x.y = sanitized(x.y, ch); x.z = sanitized(x.z, ch); x.w = sanitized(x.w, ch);
                if (ch) g4[i] = x;
            }
            s += x.x * x.x; s += x.y * x.y; s += x.z * x.z; s += x.w * x.w;
        };
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
            one(i, a); one(i + stride, b); one(i + 2 * stride, c); one(i + 3 * stride, d);
        }
        for (; i < n4; i += stride) one(i, g4[i]);
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {       // the last n % 4 values
            const long k = (n4 << 2) + threadIdx.x;
            bool ch = false;
            const float x = sanitize ? sanitized(g[k], ch) : g[k];
            if (ch) g[k] = x;
            s += x * x;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            bool ch = false;
            const float x = sanitize ? sanitized(g[i], ch) : g[i];
            if (ch) g[i] = x;
            s += x * x;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void sum_partial_kernel(const float* __restrict__ partial, int n, float* __restrict__ out,
                                                         int accumulate) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *out = accumulate ? *out + s : s;
}

}  // namespace

#define DISPATCH_DTYPE(dtype, name, ...)                                  \
    if ((dtype) == M324_BF16) { using T = bf16_t; __VA_ARGS__; }          \
    else if ((dtype) == M324_F32) { using T = float; __VA_ARGS__; }       \
    else M324_FAIL(M324_ERR_UNSUPPORTED, name ": dtype %d", (int)(dtype))

static int grid_for(long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

extern "C" int m324_cast(const void* in, long ld_in, int in_dtype, void* out, long ld_out, int out_dtype, int rows, int cols,
                         void* stream) {
    M324_REQUIRE(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= cols, "m324_cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const int nb = grid_for(vec ? (long)rows * (cols / 8) : (long)rows * cols);
#define M324_CAST(TI, TO)                                                                                                          \
    do {                                                                                                                           \
        if (vec) hipLaunchKernelGGL((cast8_kernel<TI, TO>), dim3(nb), dim3(256), 0, s, (const TI*)in, ld_in, (TO*)out, ld_out, rows, cols / 8); \
        else hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(nb), dim3(256), 0, s, (const TI*)in, ld_in, (TO*)out, ld_out, rows, cols);          \
    } while (0)
    if (in_dtype == M324_F32 && out_dtype == M324_BF16) M324_CAST(float, bf16_t);
    else if (in_dtype == M324_BF16 && out_dtype == M324_F32) M324_CAST(bf16_t, float);
    else if (in_dtype == M324_F32 && out_dtype == M324_F32) M324_CAST(float, float);
    else if (in_dtype == M324_BF16 && out_dtype == M324_BF16) M324_CAST(bf16_t, bf16_t);
    else
        M324_FAIL(M324_ERR_UNSUPPORTED, "m324_cast: dtypes %d -> %d", in_dtype, out_dtype);
#undef M324_CAST
    M324_CHECK_LAUNCH("m324_cast");
    return M324_OK;
}

extern "C" int m324_attention_delta(const void* O, const void* dO, long ld, float* D, int B, int H, int L, int dtype, void* stream) {
    M324_REQUIRE(O && dO && D && B > 0 && H > 0 && L > 0 && ld >= (long)H * 64 && ld % 8 == 0, "m324_attention_delta: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long waves = ceil_div((long)B * L * H, 8l);
    DISPATCH_DTYPE(dtype, "m324_attention_delta",
                   hipLaunchKernelGGL(attn_delta_kernel<T>, dim3(ceil_div(waves, 4)), dim3(256), 0, s, (const T*)O, (const T*)dO, ld, D,
                                      B, H, L));
    M324_CHECK_LAUNCH("m324_attention_delta");
    return M324_OK;
}

extern "C" int m324_attention_bwd(const void* Qs, long q_bstride, const void* K, const void* V, const void* dO, const float* lse,
                                  const float* D, void* dQ, void* dK, void* dV, int B, int H, int Lq, int Lk, float scale,
                                  int dtype, void* stream) {
    M324_REQUIRE(Qs && K && V && dO && lse && D && dQ && dK && dV, "m324_attention_bwd: null pointer");
    M324_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && H <= 65535 && B <= 65535, "m324_attention_bwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_attention_bwd", {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(ceil_div(Lq, 4), H, B), dim3(256), 0, s, (const T*)Qs, q_bstride, (const T*)K,
                           (const T*)V, (const T*)dO, lse, D, (T*)dQ, H, Lq, Lk, scale);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<T>, dim3(ceil_div(Lk, 4), H, B), dim3(256), 0, s, (const T*)Qs, q_bstride,
                           (const T*)K, (const T*)V, (const T*)dO, lse, D, (T*)dK, (T*)dV, H, Lq, Lk);
    });
    M324_CHECK_LAUNCH("m324_attention_bwd");
    return M324_OK;
}

extern "C" int m324_qkv_split_bwd(const void* dQ, const void* dK, const void* dV, const void* q_raw, long ldq, const void* k_raw,
                                  long ldk, const float* q_w, const float* k_w, float eps, void* dq_out, long ldoq, void* dk_out,
                                  long ldok, void* dv_out, long ldov, float* partial, int n_partial, int B, int L, int H, int dtype,
                                  void* stream) {
    M324_REQUIRE(partial && n_partial > 0 && n_partial <= 2048 && B > 0 && L > 0 && H > 0, "m324_qkv_split_bwd: bad arguments");
    M324_REQUIRE((!dQ || (dq_out && (!q_w || q_raw))) && (!dK || (dk_out && (!k_w || k_raw))) && (!dV || dv_out),
                 "m324_qkv_split_bwd: missing buffer");
    M324_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldoq % 8 == 0 && ldok % 8 == 0 && ldov % 8 == 0,
                 "m324_qkv_split_bwd: leading dimensions must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_qkv_split_bwd",
                   hipLaunchKernelGGL(qkv_split_bwd_kernel<T>, dim3(n_partial), dim3(256), 0, s, (const T*)dQ, (const T*)dK,
                                      (const T*)dV, (const T*)q_raw, ldq, (const T*)k_raw, ldk, q_w, k_w, eps, (T*)dq_out, ldoq,
                                      (T*)dk_out, ldok, (T*)dv_out, ldov, partial, B, L, H));
    M324_CHECK_LAUNCH("m324_qkv_split_bwd");
    return M324_OK;
}

extern "C" int m324_linear_n3_bwd(const void* A, long lda, const float* W, const float* dout, void* dA, long ldda, float* partial,
                                  int n_partial, int M, int K, int dtype, const void* mul, long ldmul, void* stream) {
    M324_REQUIRE(A && W && dout && dA && partial && n_partial > 0 && M > 0 && K > 0 && (!mul || ldmul >= K), "m324_linear_n3_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int esz = dtype == M324_BF16 ? 2 : 4;
    if (K % 8 == 0 && K >= 64 && K <= 2048 && (lda * esz) % 16 == 0 && (ldda * esz) % 16 == 0 && ((uintptr_t)A % 16) == 0 &&
        ((uintptr_t)dA % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)partial % 16) == 0 &&
        (!mul || (((uintptr_t)mul % 16) == 0 && (ldmul * esz) % 16 == 0))) {
        const int nrs = 256 / (K / 8);
        const size_t lds = (size_t)(nrs - 1) * 3 * K * sizeof(float);
        DISPATCH_DTYPE(dtype, "m324_linear_n3_bwd",
                       hipLaunchKernelGGL(linear_n3_bwd_vec_kernel<T>, dim3(n_partial), dim3(256), lds, s, (const T*)A, lda, W, dout,
                                          (T*)dA, ldda, partial, M, K, (const T*)mul, ldmul));
        M324_CHECK_LAUNCH("m324_linear_n3_bwd");
        return M324_OK;
    }
    DISPATCH_DTYPE(dtype, "m324_linear_n3_bwd",
                   hipLaunchKernelGGL(linear_n3_bwd_kernel<T>, dim3(n_partial), dim3(256), 0, s, (const T*)A, lda, W, dout, (T*)dA,
                                      ldda, partial, M, K, (const T*)mul, ldmul));
    M324_CHECK_LAUNCH("m324_linear_n3_bwd");
    return M324_OK;
}

extern "C" int m324_mse_bwd(const float* pred, const float* target, const float* grad_scale, float coef, float* d, long n,
                            void* stream) {
    M324_REQUIRE(pred && target && d && n > 0, "m324_mse_bwd: bad arguments");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, grad_scale, coef, d, n);
    M324_CHECK_LAUNCH("m324_mse_bwd");
    return M324_OK;
}

extern "C" int m324_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, const float* grad_scale, void* stream) {
    M324_REQUIRE(p && g && m && v && n > 0 && step >= 1, "m324_adamw: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, grad_scale);
    M324_CHECK_LAUNCH("m324_adamw");
    return M324_OK;
}

extern "C" int m324_adamw_flat(float* p, const float* g, float* m, float* v, long n, long n_decay, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, const float* grad_scale, void* stream) {
    M324_REQUIRE(p && g && m && v && n > 0 && step >= 1 && n_decay >= 0 && n_decay <= n, "m324_adamw_flat: bad arguments");
    M324_REQUIRE(n % 4 == 0 && n_decay % 4 == 0 && ((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                     ((uintptr_t)v % 16) == 0,
                 "m324_adamw_flat: the flat buffers must be 16-byte aligned with n and n_decay multiples of 4");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g,
                       (float4*)m, (float4*)v, n / 4, n_decay / 4, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    M324_CHECK_LAUNCH("m324_adamw_flat");
    return M324_OK;
}

extern "C" int m324_weight_mirror(const float* src, void* dst, void* dstT, const m324_mirror_item* items_dev, int n_items, long n_tiles,
                                  void* stream) {
    M324_REQUIRE(src && dst && items_dev && n_items > 0 && n_tiles > 0 && n_tiles <= 0x7FFFFFFFL, "m324_weight_mirror: bad arguments");
    M324_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0 && ((uintptr_t)dstT % 4) == 0, "m324_weight_mirror: unaligned buffers");
    hipLaunchKernelGGL(weight_mirror_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (bf16_t*)dstT,
                       items_dev, n_items);
    M324_CHECK_LAUNCH("m324_weight_mirror");
    return M324_OK;
}

extern "C" int m324_grad_sumsq(float* g, long n, int sanitize, float* partial, float* out, int accumulate, void* stream) {
    M324_REQUIRE(g && partial && out && n > 0, "m324_grad_sumsq: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = ((uintptr_t)g % 16) == 0 && n >= 4096;
    const int nb = vec ? (grid_for(n / 16) < 1024 ? grid_for(n / 16) : 1024) : (grid_for(n) < 1024 ? grid_for(n) : 1024);
    if (vec) hipLaunchKernelGGL(grad_sanitize_sumsq_kernel<true>, dim3(nb), dim3(256), 0, s, g, n, sanitize, partial);
    else hipLaunchKernelGGL(grad_sanitize_sumsq_kernel<false>, dim3(nb), dim3(256), 0, s, g, n, sanitize, partial);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(1), dim3(64), 0, s, partial, nb, out, accumulate);
    M324_CHECK_LAUNCH("m324_grad_sumsq");
    return M324_OK;
}

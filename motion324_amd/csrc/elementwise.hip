// HBM-bound kernels of the Motion324 hot path: normalisation, layout changes, patchify, point
// Fourier features, token assembly, the 3-wide regression head and the MSE loss.
// One wave (64 lanes) per token row wherever a row reduction is needed; 16-byte accesses per lane.
#include "common.h"

namespace {

constexpr int LN_MAXV = 4;   // float4 per lane -> C <= 1024

__device__ __forceinline__ long remap_row(long r, int gin, int gout, int off) {
    return gin > 0 ? (r / gin) * gout + (r % gin) + off : r;
}

// mean / rstd of a row held as nv float4 per lane (two-pass, like torch's LayerNorm)
// slot i of a lane covers columns lane*4 + i*256 .. +3; slots past C are skipped (statically unrolled
// so the row stays in registers)
#define LN_FOR(i, c) _Pragma("unroll") for (int i = 0, c = lane * 4; i < LN_MAXV; ++i, c += 256) if (c < C)

__device__ __forceinline__ void row_load(float4 (&v)[LN_MAXV], const float* src, int lane, int C) {
    LN_FOR(i, c) v[i] = *reinterpret_cast<const float4*>(src + c);
}
__device__ __forceinline__ void row_stats(const float4 (&v)[LN_MAXV], int lane, int C, float eps, float& mean, float& rstd) {
    float s = 0.f;
    LN_FOR(i, c) s += v[i].x + v[i].y + v[i].z + v[i].w;
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
    LN_FOR(i, c) {
        float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += a * a + b * b + cc * cc + d * d;
    }
    rstd = rsqrtf(wave_sum(q) / (float)C + eps);
}

template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

// ----------------------------------------------------------------------------------- layernorm
// A wave normalises TWO rows at a time (rows 2w, 2w + 1 of the workgroup's eight): the pass is a chain of dependent steps
// -- load, mean butterfly, variance butterfly, store -- and two independent chains per wave overlap each other's shuffle
// and memory latency (same two-pass statistics per row as before: bit-identical results).
template <typename T, typename TX = float>
__global__ __launch_bounds__(256) void layernorm1_kernel(const TX* __restrict__ x, long ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, float eps, T* __restrict__ y, long ldy,
                                                         int rows, int C, int gin, int gout, int off) {   // one row per wave (M324_LN_ROWS=1: A/B)
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TX* xr = x + remap_row(row, gin, gout, off) * ldx;
    float4 v[LN_MAXV];
    LN_FOR(i, c) v[i] = load4<TX>(xr + c);
    float mean, rstd;
    row_stats(v, lane, C, eps, mean, rstd);
    T* yr = y + row * ldy;
    LN_FOR(i, c) {
        float4 ww = *reinterpret_cast<const float4*>(w + c);
        float4 bb = b ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        store4<T>(yr + c, (v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                  (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
    }
}

template <typename T, typename TX = float>
__device__ __forceinline__ void ln_rows2(const TX* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ b,
                                         float eps, T* __restrict__ y, long ldy, int rows, int C, int gin, int gout, int off, int block) {
    const int lane = threadIdx.x & 63;
    const long row0 = (long)block * 8 + (threadIdx.x >> 6) * 2;
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;
    const long row1 = two ? row0 + 1 : row0;                 // a lone last row is simply done twice (same values stored twice)
    const TX* xa = x + remap_row(row0, gin, gout, off) * ldx;
    const TX* xb = x + remap_row(row1, gin, gout, off) * ldx;
    float4 va[LN_MAXV], vb[LN_MAXV];
    LN_FOR(i, c) { va[i] = load4<TX>(xa + c); vb[i] = load4<TX>(xb + c); }
    float sa = 0.f, sb = 0.f;
    LN_FOR(i, c) { sa += va[i].x + va[i].y + va[i].z + va[i].w; sb += vb[i].x + vb[i].y + vb[i].z + vb[i].w; }
    wave_sum2(sa, sb);
    const float ma = sa / (float)C, mb = sb / (float)C;
    float qa = 0.f, qb = 0.f;
    LN_FOR(i, c) {
        float p = va[i].x - ma, q = va[i].y - ma, r = va[i].z - ma, t = va[i].w - ma;
        qa += p * p + q * q + r * r + t * t;
        p = vb[i].x - mb, q = vb[i].y - mb, r = vb[i].z - mb, t = vb[i].w - mb;
        qb += p * p + q * q + r * r + t * t;
    }
    wave_sum2(qa, qb);
    const float ra = rsqrtf(qa / (float)C + eps), rb = rsqrtf(qb / (float)C + eps);
    T* ya = y + row0 * ldy;
    T* yb = y + row1 * ldy;
    LN_FOR(i, c) {
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        const float4 bb = b ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        store4<T>(ya + c, (va[i].x - ma) * ra * ww.x + bb.x, (va[i].y - ma) * ra * ww.y + bb.y,
                  (va[i].z - ma) * ra * ww.z + bb.z, (va[i].w - ma) * ra * ww.w + bb.w);
        store4<T>(yb + c, (vb[i].x - mb) * rb * ww.x + bb.x, (vb[i].y - mb) * rb * ww.y + bb.y,
                  (vb[i].z - mb) * rb * ww.z + bb.z, (vb[i].w - mb) * rb * ww.w + bb.w);
    }
}

template <typename T, typename TX = float>
__global__ __launch_bounds__(256) void layernorm_kernel(const TX* __restrict__ x, long ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, T* __restrict__ y, long ldy,
                                                        int rows, int C, int gin, int gout, int off) {
    ln_rows2<T, TX>(x, ldx, w, b, eps, y, ldy, rows, C, gin, gout, off, (int)blockIdx.x);
}

// Two LayerNorms of the same width in one launch (m324_layernorm_pair: the decoder's norm_q and norm_kv, 2048 rows each -- two
// launches of a few microseconds whose cost is the launch).
struct LnProblem {
    const float* x; long ldx;
    const float* w; const float* b; float eps;
    void* y; long ldy;
    int rows, gin, gout, off;
};
template <typename T>
__global__ __launch_bounds__(256) void layernorm_pair_kernel(LnProblem p0, LnProblem p1, int C, int blocks0) {
    if ((int)blockIdx.x < blocks0)
        ln_rows2<T, float>(p0.x, p0.ldx, p0.w, p0.b, p0.eps, (T*)p0.y, p0.ldy, p0.rows, C, p0.gin, p0.gout, p0.off, (int)blockIdx.x);
    else
        ln_rows2<T, float>(p1.x, p1.ldx, p1.w, p1.b, p1.eps, (T*)p1.y, p1.ldy, p1.rows, C, p1.gin, p1.gout, p1.off, (int)blockIdx.x - blocks0);
}

// ----------------------------------------------------------------------------------- LayerNorm fold: row statistics
// part[cb][m] = (sum, M2) of the 64 columns of block cb (left by a producer GEMM's epilogue) -> rowstat[m] = (rstd, -rstd mean).
// Chan's parallel update: M2 = sum_b M2_b + 64 (mean_b - mean)^2 -- no E[x^2] - mean^2 cancellation.  Thread per row,
// consecutive threads read consecutive float2 of a block (coalesced).
__global__ __launch_bounds__(256) void rowstats_finish_kernel(const float2* __restrict__ part, int ncb, int M, float eps,
                                                              float2* __restrict__ rowstat) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    if ((ncb & 1) == 0) {
        // even block counts: the arithmetic of the consumer GEMMs that merge the table themselves (gemm_tile.h ln_chan_step /
        // ln_combine_halves: each half of the blocks by Chan's update, block after block, then the pairwise form for equal counts) --
        // a folded product does not depend on WHO merged its rows' statistics (round 5: schedule v15 takes the merged table)
        const int h = ncb >> 1;
        float mean[2], m2[2];
        for (int half = 0; half < 2; ++half) {
            const float2 p0 = part[(long)(half * h) * M + m];
            mean[half] = p0.x * (1.0f / 64.0f), m2[half] = p0.y;
            for (int i = 1; i < h; ++i) {
                const float2 pb = part[(long)(half * h + i) * M + m];
                const float d = pb.x * (1.0f / 64.0f) - mean[half];
                mean[half] = fmaf(d, 1.0f / (float)(i + 1), mean[half]);
                m2[half] = m2[half] + fmaf(d * d, 64.0f * (float)i / (float)(i + 1), pb.y);
            }
        }
        const float d = mean[1] - mean[0];
        const float mm2 = (m2[0] + m2[1]) + (d * d) * (32.0f * (float)h);
        const float mean_ = fmaf(0.5f, d, mean[0]);
        const float rstd = rsqrtf(mm2 / (64.0f * (float)ncb) + eps);
        rowstat[m] = make_float2(rstd, -rstd * mean_);
        return;
    }
    float s = 0.f;
    for (int b = 0; b < ncb; ++b) s += part[(long)b * M + m].x;
    const float n = 64.0f * (float)ncb, mean = s / n;
    float m2 = 0.f;
    for (int b = 0; b < ncb; ++b) {
        const float2 pb = part[(long)b * M + m];
        const float d = pb.x * (1.0f / 64.0f) - mean;
        m2 += fmaf(64.0f * d, d, pb.y);
    }
    const float rstd = rsqrtf(m2 / n + eps);
    rowstat[m] = make_float2(rstd, -rstd * mean);
}

// The same table from an fp32 stream (head of a chain), two-pass like layernorm_kernel, plus the bf16 copy of the row.
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, long ldx, int rows, int C, float eps,
                                                       float2* __restrict__ rowstat, bf16_t* __restrict__ copy, long ldcopy) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[LN_MAXV];
    row_load(v, x + row * ldx, lane, C);
    float mean, rstd;
    row_stats(v, lane, C, eps, mean, rstd);
    if (lane == 0) rowstat[row] = make_float2(rstd, -rstd * mean);
    if (copy) {
        LN_FOR(i, c) store4<bf16_t>(copy + row * ldcopy + c, v[i].x, v[i].y, v[i].z, v[i].w);
    }
}

// ----------------------------------------------------------------------------------- qkv split
// One workgroup per (64-token tile, head, batch).  Thread t: token = t >> 2, 16 columns (t & 3) * 16.
// Each of the three sources can be emitted row-major (head-major [B,H,L,64]) and / or transposed
// ([B,H,64,Lp], key quarters of every 16-token group in the order 0,2,1,3, zero padded): inference needs
// Q, K, Vt; the attention backward additionally needs V, Qt, Kt (and dO / dOt, produced by the same kernel).
template <typename T>
__global__ __launch_bounds__(256) void qkv_split_kernel(const T* __restrict__ qs, long ldq, const T* __restrict__ ks, long ldk,
                                                        const T* __restrict__ vs, long ldv, const float* __restrict__ qw,
                                                        const float* __restrict__ kw, float eps, float q_scale,
                                                        T* __restrict__ Q, T* __restrict__ K, T* __restrict__ V,
                                                        T* __restrict__ Qt, T* __restrict__ Kt, T* __restrict__ Vt, int L, int H,
                                                        int Lp) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x, tok = t >> 2, part = t & 3;
    const int b = blockIdx.z, h = blockIdx.y, l0 = blockIdx.x * 64;
    const int l = l0 + tok;
    const bool ok = l < L;
    const long srow = (long)b * L + l;
    const long hbase = ((long)b * H + h);

    auto one = [&](const T* src, long ld, const float* w, float post, T* dst, T* dstT) {
        float v[16];
        if (ok) {                                 // 16-byte accesses: an 8-byte piece of a line costs a transaction like 16 do
            const T* p = src + srow * ld + h * 64 + part * 16;
            float lo[8], hi8[8];
            load8(p, lo);
            load8(p + 8, hi8);
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = lo[i]; v[8 + i] = hi8[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = 0.f;
        }
        if (w) {   // RMSNorm over the 64 columns of this (token, head): 4 neighbouring lanes
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) ss += v[i] * v[i];
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const float r = rsqrtf(ss * (1.0f / 64.0f) + eps);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = v[i] * r * w[part * 16 + i];
        }
        if (post != 1.0f) {   // softmax scale * log2(e) folded into Q before its (single) rounding
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] *= post;
        }
        if (dst && ok) {
            T* o = dst + (hbase * L + l) * 64 + part * 16;
            const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
            const float hi8[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
            store8(o, lo);
            store8(o + 8, hi8);
        }
        if (dstT) {
            __syncthreads();   // previous user of the tile is done
#pragma unroll
            for (int i = 0; i < 16; ++i) tile[part * 16 + i][tok] = v[i];     // zeros beyond L: the padding
            __syncthreads();
            const int d = t >> 2;   // output row d, 16 tokens (t & 3) * 16
            T* o = dstT + (hbase * 64 + d) * (long)Lp + l0 + part * 16;
            // token quarters of the 16-group go out in the order 0, 2, 1, 3: two 16-byte chunks (q0 | q2) and (q1 | q3)
            const float* tr = &tile[d][part * 16];
            const float c0[8] = {tr[0], tr[1], tr[2], tr[3], tr[8], tr[9], tr[10], tr[11]};
            const float c1[8] = {tr[4], tr[5], tr[6], tr[7], tr[12], tr[13], tr[14], tr[15]};
            store8(o, c0);
            store8(o + 8, c1);
        }
    };
    if (qs) one(qs, ldq, qw, q_scale, Q, Qt);
    if (ks) one(ks, ldk, kw, 1.0f, K, Kt);
    if (vs) one(vs, ldv, nullptr, 1.0f, V, Vt);
}

// ----------------------------------------------------------------------------------- patchify
// One thread per output pixel (frame, oy, ox): 4 bilinear taps x 3 interleaved channels.
// PyTorch upsample_bilinear2d, align_corners=False: src = scale * (dst + 0.5) - 0.5 clamped at 0,
// scale = in / out; i1 = min(i0 + 1, in - 1).
// TIN = unsigned char: byte frames, every tap converted as (float)v / 255.0f -- the caller's `frames.float() / 255.0`, IEEE division.
template <typename TIN>
__device__ __forceinline__ float patch_tap(const TIN* p, int c) {
    if constexpr (sizeof(TIN) == 1) return (float)p[c] / 255.0f;
    else return p[c];
}
template <typename T, typename TIN>
__global__ __launch_bounds__(256) void patchify_kernel(const TIN* __restrict__ video, int F, int Hin, int Win, int size,
                                                       int patch, T* __restrict__ out, int Kp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)F * size * size;
    if (idx >= total) return;
    const int ox = (int)(idx % size), oy = (int)((idx / size) % size), f = (int)(idx / ((long)size * size));
    const float sh = (float)Hin / (float)size, sw = (float)Win / (float)size;
    float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)fy, Hin - 1), x0 = min((int)fx, Win - 1);
    const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const TIN* base = video + (long)f * Hin * Win * 3;
    const TIN* p00 = base + ((long)y0 * Win + x0) * 3;
    const TIN* p01 = base + ((long)y0 * Win + x1) * 3;
    const TIN* p10 = base + ((long)y1 * Win + x0) * 3;
    const TIN* p11 = base + ((long)y1 * Win + x1) * 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};   // dinov2.py:7-8
    const int g = size / patch;
    const int py = oy / patch, ky = oy % patch, px = ox / patch, kx = ox % patch;
    T* orow = out + ((long)f * g * g + (long)py * g + px) * Kp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = hy * (hx * patch_tap(p00, c) + lx * patch_tap(p01, c)) + ly * (hx * patch_tap(p10, c) + lx * patch_tap(p11, c));
        Elem<T>::store(orow + c * patch * patch + ky * patch + kx, (v - mean[c]) / stdv[c]);
    }
    // zero the K padding once per patch row (the thread of the patch's first pixel)
    if (ky == 0 && kx == 0)
        for (int c = 3 * patch * patch; c < Kp; ++c) Elem<T>::store(orow + c, 0.f);
}

// ----------------------------------------------------------------------------------- point features
// 64 columns per point: [sin(x e_j), sin(y e_j), sin(z e_j) | cos ... | x y z | 0 x 13], e_j = 2^j pi.
template <typename T>
__global__ __launch_bounds__(256) void point_encode_kernel(const float* __restrict__ xyz, int P, T* __restrict__ out, long ld) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per (point, column)
    const long p = idx >> 6;
    const int c = (int)(idx & 63);
    if (p >= P) return;
    float v = 0.f;
    if (c < 48) {
        const int cc = c % 24, axis = cc >> 3, j = cc & 7;
        // the reference multiplies by the fp32 constant fl(2^j * pi) (Pcd_motion.py:164): reproduce it
        const float e = (float)(1 << j) * 3.14159274101257324f;
        const float arg = xyz[p * 3 + axis] * e;
        v = c < 24 ? sinf(arg) : cosf(arg);
    } else if (c < 51) {
        v = xyz[p * 3 + (c - 48)];
    }
    Elem<T>::store(out + p * ld + c, v);
}

template <typename T>
__global__ __launch_bounds__(256) void point_concat_kernel(const float* __restrict__ normal, const float* __restrict__ rgb,
                                                           int P, T* __restrict__ feat, int C, int Kp) {
    const int w = Kp - C;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long p = idx / w;
    const int c = (int)(idx % w);
    if (p >= P) return;
    float v = 0.f;
    if (c < 3) v = normal[p * 3 + c];
    else if (c < 6) v = rgb[p * 3 + c - 3];
    Elem<T>::store(feat + p * Kp + C + c, v);
}

__global__ void dino_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos0, float* __restrict__ x, int F,
                                int rpf, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)F * C) return;
    const int c = (int)(idx % C);
    const long f = idx / C;
    x[f * rpf * C + c] = cls[c] + pos0[c];
}

// ----------------------------------------------------------------------------------- token assembly
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ dino_x, const float* __restrict__ dw,
                                                       const float* __restrict__ db, float eps_d, const float* __restrict__ pos,
                                                       const float* __restrict__ sp0, const float* __restrict__ spr,
                                                       const float* __restrict__ mesh, const float* __restrict__ lw, float eps_in,
                                                       float* __restrict__ out, int B, int T, int K, int P, int C,
                                                       unsigned drop_thr, float drop_scale, unsigned long long drop_seed) {
    const int lane = threadIdx.x & 63;
    const int Lt = 4 + K + P;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T * Lt) return;
    const int j = (int)(row % Lt);
    const int t = (int)((row / Lt) % T);
    const int b = (int)(row / ((long)Lt * T));
    float4 v[LN_MAXV];
    if (j < 4 + K) {   // wave-uniform: a wave owns one row
        const float* src = j < 4 ? (t == 0 ? sp0 : spr) + (long)j * C : mesh + ((long)b * K + (j - 4)) * C;
        row_load(v, src, lane, C);
    } else {
        const int p = j - 4 - K;
        row_load(v, dino_x + (((long)b * T + t) * (P + 1) + 1 + p) * C, lane, C);
        float mean, rstd;
        row_stats(v, lane, C, eps_d, mean, rstd);
        const float* pe = pos + ((long)t * P + p) * C;
        LN_FOR(i, c) {
            float4 ww = *reinterpret_cast<const float4*>(dw + c), bb = *reinterpret_cast<const float4*>(db + c);
            float4 pp = *reinterpret_cast<const float4*>(pe + c);
            v[i].x = (v[i].x - mean) * rstd * ww.x + bb.x + pp.x;
            v[i].y = (v[i].y - mean) * rstd * ww.y + bb.y + pp.y;
            v[i].z = (v[i].z - mean) * rstd * ww.z + bb.z + pp.z;
            v[i].w = (v[i].w - mean) * rstd * ww.w + bb.w + pp.w;
        }
        if (drop_thr) {   // pos_drop: counter-based mask, one SplitMix64 draw per element of x[B, T*P, C]
            const unsigned long long base = ((((unsigned long long)b * T + t) * P) + p) * (unsigned long long)C;
            LN_FOR(i, c) {
                float* e = reinterpret_cast<float*>(&v[i]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    e[q] = (unsigned)(splitmix64((base + c + q) * 0xD1342543DE82EF95ull + drop_seed) >> 40) >= drop_thr
                               ? e[q] * drop_scale : 0.f;
            }
        }
    }
    float* o = out + row * C;
    if (!lw) {   // training: the un-normalised concatenation is needed by the input-LayerNorm backward
        LN_FOR(i, c) *reinterpret_cast<float4*>(o + c) = v[i];
        return;
    }
    float mean, rstd;
    row_stats(v, lane, C, eps_in, mean, rstd);
    LN_FOR(i, c) {
        float4 ww = *reinterpret_cast<const float4*>(lw + c);
        *reinterpret_cast<float4*>(o + c) = make_float4((v[i].x - mean) * rstd * ww.x, (v[i].y - mean) * rstd * ww.y,
                                                        (v[i].z - mean) * rstd * ww.z, (v[i].w - mean) * rstd * ww.w);
    }
}

// ----------------------------------------------------------------------------------- 3-wide head
template <typename T>
__global__ __launch_bounds__(256) void linear_n3_kernel(const T* __restrict__ A, long lda, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out, int M, int K) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* a = A + row * lda;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < K; c += 256) {
        float4 x = load4<T>(a + c);
        float4 w0 = *reinterpret_cast<const float4*>(W + c), w1 = *reinterpret_cast<const float4*>(W + K + c),
               w2 = *reinterpret_cast<const float4*>(W + 2 * (long)K + c);
        s0 += x.x * w0.x + x.y * w0.y + x.z * w0.z + x.w * w0.w;
        s1 += x.x * w1.x + x.y * w1.y + x.z * w1.z + x.w * w1.w;
        s2 += x.x * w2.x + x.y * w2.y + x.z * w2.z + x.w * w2.w;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        out[row * 3 + 0] = s0 + bias[0];
        out[row * 3 + 1] = s1 + bias[1];
        out[row * 3 + 2] = s2 + bias[2];
    }
}

// ----------------------------------------------------------------------------------- MSE
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s += d * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void mse_final_kernel(const float* __restrict__ partial, int np, long n, float weight,
                                                       float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *out = weight * (s / (float)n);
}


// ----------------------------------------------------------------------------------- trajectory smoothing
// Post-processing of the predicted trajectories (reference utils/inference_utils.py:99-148, a CPU triple loop
// over B*N*3): pass 1 freezes points whose frame-to-frame displacement (of the ORIGINAL trajectory) is below
// the threshold -- sequential in t because frame t copies the already-smoothed frame t-1; pass 2 is
// scipy.ndimage.gaussian_filter1d along t (radius int(4 sigma + 0.5), weights exp(-x^2 / 2 sigma^2)
// normalised, mode='nearest' = clamped indices).  One thread per (batch, point): fully coalesced over n.
__global__ __launch_bounds__(256) void smooth_threshold_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                               int T, int N, float thr) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * N) return;
    const int b = (int)(idx / N), n = (int)(idx % N);
    const float* p = in + ((long)b * T * N + n) * 3;
    float* q = out + ((long)b * T * N + n) * 3;
    float ox = p[0], oy = p[1], oz = p[2];          // original frame t-1
    float sx = ox, sy = oy, sz = oz;                // smoothed frame t-1
    q[0] = sx; q[1] = sy; q[2] = sz;
    for (int t = 1; t < T; ++t) {
        const long o = (long)t * N * 3;
        const float x = p[o], y = p[o + 1], z = p[o + 2];
        const float dx = x - ox, dy = y - oy, dz = z - oz;
        const bool still = thr >= 0.f && sqrtf(dx * dx + dy * dy + dz * dz) < thr;
        if (!still) { sx = x; sy = y; sz = z; }
        q[o] = sx; q[o + 1] = sy; q[o + 2] = sz;
        ox = x; oy = y; oz = z;
    }
}

__global__ __launch_bounds__(256) void smooth_gauss_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T,
                                                           int N, float sigma, int radius) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one thread per (b, t, n)
    if (idx >= (long)B * T * N) return;
    const int n = (int)(idx % N), t = (int)((idx / N) % T), b = (int)(idx / ((long)N * T));
    float wsum = 0.f;
    for (int r = -radius; r <= radius; ++r) wsum += expf(-0.5f * (float)(r * r) / (sigma * sigma));
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int r = -radius; r <= radius; ++r) {
        const int tt = min(max(t + r, 0), T - 1);
        const float w = expf(-0.5f * (float)(r * r) / (sigma * sigma)) / wsum;
        const float* p = in + (((long)b * T + tt) * N + n) * 3;
        ax += w * p[0]; ay += w * p[1]; az += w * p[2];
    }
    float* q = out + idx * 3;
    q[0] = ax; q[1] = ay; q[2] = az;
}

// FIR along t with clamped ('nearest') borders: scipy.signal.savgol_filter(mode='nearest') = convolve1d with the
// Savitzky-Golay coefficients (host-computed, symmetric for deriv = 0), accumulated in fp64 like scipy does.
__global__ __launch_bounds__(256) void smooth_fir_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T, int N,
                                                         const double* __restrict__ coef, int window) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // one thread per (b, t, n)
    if (idx >= (long)B * T * N) return;
    const int n = (int)(idx % N), t = (int)((idx / N) % T), b = (int)(idx / ((long)N * T));
    const int h = window / 2;
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int j = 0; j < window; ++j) {
        const int tt = min(max(t + h - j, 0), T - 1);           // convolution: y[t] = sum_j c[j] x[t + h - j]
        const double w = coef[j];
        const float* p = in + (((long)b * T + tt) * N + n) * 3;
        ax += w * (double)p[0]; ay += w * (double)p[1]; az += w * (double)p[2];
    }
    float* q = out + idx * 3;
    q[0] = (float)ax; q[1] = (float)ay; q[2] = (float)az;
}

// One Euro filter (utils/inference_utils.py:58-97,186-195): a causal adaptive low-pass per scalar coordinate, sequential in
// t.  One thread per (batch, point, xyz).  Arithmetic in fp64: under the reference's pinned numpy 1.26 the filter state
// (python floats x np.float32 scalars) is float64 from the second sample on, only the stored result is rounded to fp32.
__global__ __launch_bounds__(256) void smooth_oneeuro_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T,
                                                             int N, double mincutoff, double beta, double dcutoff) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // (b, n, c)
    if (idx >= (long)B * N * 3) return;
    const int b = (int)(idx / ((long)N * 3));
    const long nc = idx % ((long)N * 3);
    const float* p = in + (long)b * T * N * 3 + nc;
    float* q = out + (long)b * T * N * 3 + nc;
    const double two_pi = 6.283185307179586;
    const double rd = two_pi * dcutoff, alpha_d = rd / (rd + 1.0);
    double x_prev = (double)p[0], dx_prev = 0.0;
    q[0] = p[0];
    for (int t = 1; t < T; ++t) {
        const long o = (long)t * N * 3;
        const double x = (double)p[o];
        const double dx = t == 1 ? (double)(p[o] - p[0]) : x - x_prev;     // the first difference is float32 - float32
        const double dx_hat = alpha_d * dx + (1.0 - alpha_d) * dx_prev;
        const double r = two_pi * (mincutoff + beta * fabs(dx_hat));
        const double alpha = r / (r + 1.0);
        const double x_hat = alpha * x + (1.0 - alpha) * x_prev;
        x_prev = x_hat;
        dx_prev = dx_hat;
        q[o] = (float)x_hat;
    }
}

// Nearest reference point of every query point (brute force, squared distances in fp32 from fp32 differences; the lowest
// index wins ties): the vertex-colour assignment of the caller's pre-step (scripts/inference_with_video_mesh.py:112-115,
// a cKDTree query there).  Reference points are staged through LDS in tiles of 1024.
__global__ __launch_bounds__(256) void nearest_point_kernel(const float* __restrict__ query, int nq, const float* __restrict__ ref,
                                                            int nr, int* __restrict__ idx_out) {
    __shared__ float tile[1024 * 3];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const bool live = qi < nq;
    const float qx = live ? query[qi * 3] : 0.f, qy = live ? query[qi * 3 + 1] : 0.f, qz = live ? query[qi * 3 + 2] : 0.f;
    float best = INFINITY;
    int besti = 0;
    for (int r0 = 0; r0 < nr; r0 += 1024) {
        const int cnt = min(1024, nr - r0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += 256) tile[e] = ref[(long)r0 * 3 + e];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float dx = tile[j * 3] - qx, dy = tile[j * 3 + 1] - qy, dz = tile[j * 3 + 2] - qz;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; besti = r0 + j; }
        }
    }
    if (live) idx_out[qi] = besti;
}

// =====================================================================================================
// Training-side kernels (backward of the HBM-bound ops + layout helpers for the backward GEMMs)
// =====================================================================================================

// out[c][r] = in[r][c]; rows beyond `rows` up to rows_pad are written as zeros (the backward GEMMs contract
// over the token dimension, which must be a multiple of the K-tile).  64 x 64 tiles through LDS.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, long ld_in, T* __restrict__ out, long ld_out,
                                                        int rows, int cols, int rows_pad) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int id = t + 256 * i, r = id >> 6, c = id & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? Elem<T>::load(in + (long)(r0 + r) * ld_in + c0 + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int id = t + 256 * i, c = id >> 6, r = id & 63;
        if (c0 + c < cols && r0 + r < rows_pad) Elem<T>::store(out + (long)(c0 + c) * ld_out + r0 + r, tile[r][c]);
    }
}

// out[c] (+)= sum_r x[r][c] : bias gradients and the second stage of LayerNorm weight gradients.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long ld, float* __restrict__ out, int rows,
                                                     int cols, int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (int r = w; r < rows; r += 4) s += Elem<T>::load(x + (long)r * ld + c);
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        out[c] = accumulate ? out[c] + v : v;
    }
}

// many rows: blockIdx.y = row chunk, partial sums into scratch[chunk][cols] (then reduced by the wide kernel)
template <typename T>
__global__ __launch_bounds__(256) void colsum_chunk_kernel(const T* __restrict__ x, long ld, float* __restrict__ scratch, int rows,
                                                           int cols) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (int r = blockIdx.y * 4 + w; r < rows; r += gridDim.y * 4) s += Elem<T>::load(x + (long)r * ld + c);
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols)
        scratch[(long)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// few rows, very many columns (sum over the frames of a shared-query gradient): thread per column, coalesced
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* __restrict__ x, long ld, float* __restrict__ out, int rows,
                                                          long cols, int accumulate) {
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < cols; c += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += Elem<T>::load(x + (long)r * ld + c);
        out[c] = accumulate ? out[c] + s : s;
    }
}

// The two kernels above with four neighbouring columns per lane (8 / 16 bytes per access instead of 2 / 4; cols % 4 == 0, aligned rows):
// the bias-gradient sums of the training step's tall bf16 gradients (49152 x 768: 66 -> ~20 us) and the sums over the batch of a shared
// query's gradient (8 x 3.1 M: 41 -> ~12 us).  Same summation order per column as the scalar forms.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void acc4(float4& s, const float4 v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }

template <typename T>
__global__ __launch_bounds__(256) void colsum_chunk4_kernel(const T* __restrict__ x, long ld, float* __restrict__ scratch, int rows,
                                                            int cols) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, c = (blockIdx.x * 64 + lane) * 4, w = threadIdx.x >> 6;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
        const int step = gridDim.y * 4;
        int r = blockIdx.y * 4 + w;
        for (; r + 3 * step < rows; r += 4 * step) {       // four rows in flight
            const float4 a = ld4(x + (long)r * ld + c), b = ld4(x + (long)(r + step) * ld + c), d = ld4(x + (long)(r + 2 * step) * ld + c),
                         e = ld4(x + (long)(r + 3 * step) * ld + c);
            acc4(s, a); acc4(s, b); acc4(s, d); acc4(s, e);
        }
        for (; r < rows; r += step) acc4(s, ld4(x + (long)r * ld + c));
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < cols) {
        float4 v = red[0][lane];
        acc4(v, red[1][lane]); acc4(v, red[2][lane]); acc4(v, red[3][lane]);
        *reinterpret_cast<float4*>(scratch + (long)blockIdx.y * cols + c) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_wide4_kernel(const T* __restrict__ x, long ld, float* __restrict__ out, int rows,
                                                           long cols, int accumulate) {
    for (long c = ((long)blockIdx.x * 256 + threadIdx.x) * 4; c < cols; c += (long)gridDim.x * 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int r = 0; r < rows; ++r) acc4(s, ld4(x + (long)r * ld + c));
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(out + c);
            s = make_float4(o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w);
        }
        *reinterpret_cast<float4*>(out + c) = s;
    }
}

// m324_colsum_multi: one launch for many (destination, chain of fp32 row blocks) pairs.  The table travels in the kernel arguments.
// A destination's workgroups: "wide" (every source of its chain has <= 64 rows: the split-K partials of a weight gradient) -- a thread
// owns four neighbouring columns and walks the rows, then the chain, in order (colsum_wide_kernel's arithmetic); "tall" (the
// per-workgroup partials of a LayerNorm backward: few columns, hundreds of rows) -- 64 columns per workgroup, wave w of 16 takes rows
// w, w + 16, ..., the sixteen partial sums are added in wave order.
constexpr int CSM_MAX = 64, CSM_THREADS = 1024, CSM_WAVES = CSM_THREADS / 64;
struct CsmTable {
    m324_colsum_item it[CSM_MAX];
    int blk0[CSM_MAX + 1];          // first workgroup of item i's destination (heads only; a chained item repeats its head's)
    int n;
};
// 1024 threads: a tall source (up to 1024 partial rows of 64 .. 2304 columns) is walked by 16 waves at once, rows w, w + 16, ...
__global__ __launch_bounds__(CSM_THREADS) void colsum_multi_kernel(const CsmTable t) {
    __shared__ float red[CSM_WAVES][64];
    const int bid = blockIdx.x;
    int lo = 0, hi = t.n;                                    // the head whose workgroup range holds bid: blk0[lo] <= bid < blk0 of the next head
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.blk0[mid] <= bid) lo = mid; else hi = mid;
    }
    while (lo > 0 && t.it[lo].chain) --lo;                   // a chained item shares its head's range
    const int head = lo;
    int last = head;
    bool tall = t.it[head].rows > 64, vec = true;
    while (last + 1 < t.n && t.it[last + 1].chain) ++last;
    for (int i = head; i <= last; ++i) {
        tall = tall || t.it[i].rows > 64;
        vec = vec && (t.it[i].ld & 3) == 0 && (((unsigned long)t.it[i].src) & 15) == 0;
    }
    const m324_colsum_item h0 = t.it[head];
    vec = vec && (h0.cols & 3) == 0 && (((unsigned long)h0.dst) & 15) == 0;
    const int local = bid - t.blk0[head];
    if (tall) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = local * 64 + lane;
        float val = 0.f;
        for (int i = head; i <= last; ++i) {
            const m324_colsum_item it = t.it[i];
            float s = 0.f;
            if (c < it.cols) {
                const float* p = it.src + c;
#pragma unroll 4
                for (int r = w; r < it.rows; r += CSM_WAVES) s += p[(long)r * it.ld];
            }
            __syncthreads();
            red[w][lane] = s;
            __syncthreads();
            if (w == 0 && c < it.cols) {
                float v = red[0][lane];
#pragma unroll
                for (int k = 1; k < CSM_WAVES; ++k) v += red[k][lane];
                val = i == head ? (it.accumulate ? it.dst[c] + v : v) : val + v;
            }
        }
        if (w == 0 && c < h0.cols) h0.dst[c] = val;
        return;
    }
    if (vec) {
        const long c = ((long)local * CSM_THREADS + threadIdx.x) * 4;
        if (c >= h0.cols) return;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = head; i <= last; ++i) {
            const m324_colsum_item it = t.it[i];
            const float* p = it.src + c;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int r = 0; r < it.rows; ++r) {
                typedef float csm_f4 __attribute__((ext_vector_type(4)));
                const csm_f4 x = __builtin_nontemporal_load(reinterpret_cast<const csm_f4*>(p + (long)r * it.ld));     // read once
                s.x += x[0], s.y += x[1], s.z += x[2], s.w += x[3];
            }
            if (i == head) {
                if (it.accumulate) {
                    const float4 d = *reinterpret_cast<const float4*>(it.dst + c);
                    val = make_float4(d.x + s.x, d.y + s.y, d.z + s.z, d.w + s.w);
                } else {
                    val = s;
                }
            } else {
                val.x += s.x, val.y += s.y, val.z += s.z, val.w += s.w;
            }
        }
        *reinterpret_cast<float4*>(h0.dst + c) = val;
        return;
    }
    const long c = (long)local * CSM_THREADS + threadIdx.x;  // unaligned operands: one column per thread
    if (c >= h0.cols) return;
    float val = 0.f;
    for (int i = head; i <= last; ++i) {
        const m324_colsum_item it = t.it[i];
        float s = 0.f;
        for (int r = 0; r < it.rows; ++r) s += it.src[(long)r * it.ld + c];
        val = i == head ? (it.accumulate ? it.dst[c] + s : s) : val + s;
    }
    h0.dst[c] = val;
}

// GELU forward on a stored pre-activation and its backward: dz = dh * (Phi(z) + z * phi(z)).
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ z, T* __restrict__ h, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        Elem<T>::store(h + i, gelu_erf(Elem<T>::load(z + i)));
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dh, T* __restrict__ dz,
                                                       long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = Elem<T>::load(z + i);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
        Elem<T>::store(dz + i, Elem<T>::load(dh + i) * (cdf + x * pdf));
    }
}
// n % 8 == 0, 16-byte aligned operands: eight values per thread
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd8_kernel(const T* __restrict__ z, T* __restrict__ h, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8];
        load8(z + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
        store8(h + i * 8, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd8_kernel(const T* __restrict__ z, const T* __restrict__ dh, T* __restrict__ dz,
                                                        long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float x[8], g[8];
        load8(z + i * 8, x);
        load8(dh + i * 8, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float cdf = 0.5f * (1.0f + erff(x[k] * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * x[k] * x[k]);
            g[k] *= cdf + x[k] * pdf;
        }
        store8(dz + i * 8, g);
    }
}

// LayerNorm backward, one wave per row (grid-stride over rows), 8 waves per workgroup:
//   xh = (x - mean) * rstd ; g = dy * w ; dx += rstd * (g - mean(g) - xh * mean(g * xh))
//   partial[workgroup][0:C] = sum of dy * xh (dw), partial[workgroup][C:2C] = sum of dy (db) over the workgroup's rows (its 8 waves'
//   sums added in wave order through LDS)  -> summed by m324_colsum (fixed order = deterministic).  Up to 512 workgroups = 4096
//   waves: the row loop is a chain of dependent loads, and the 1024 waves of the first version left it latency-bound (110 us for
//   24 672 rows x 768; the per-wave partials then cost m324_colsum another 58 us).
constexpr int LNB_WAVES = 8;
// CAST: the bf16 copy of the resulting dx row (what the next backward GEMMs read: saves the cast pass that followed every call) and,
// as a third block of the partials, the column sums of that ROUNDED copy (the bias gradient of the Linear in front: saves its
// m324_colsum pass over the tensor).
template <typename T, bool CAST>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                            float eps, const T* __restrict__ dy, long ldy,
                                                            float* __restrict__ dx, long lddx, int accumulate,
                                                            float* __restrict__ partial, int rows, int C, int gin, int gout,
                                                            int off, bf16_t* __restrict__ dxc, long ldc) {
    constexpr int NB = CAST ? 3 : 2;
    extern __shared__ float lnb_red[];                        // [LNB_WAVES][2 C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * LNB_WAVES + wave, nw = gridDim.x * LNB_WAVES;
    float4 pw[LN_MAXV], pb[LN_MAXV], pc[CAST ? LN_MAXV : 1];
    LN_FOR(i, c) { pw[i] = make_float4(0.f, 0.f, 0.f, 0.f); pb[i] = make_float4(0.f, 0.f, 0.f, 0.f); if (CAST) pc[i] = pw[i]; }
    for (long row = gw; row < rows; row += nw) {
        const long xr = remap_row(row, gin, gout, off);
        float4 v[LN_MAXV];
        row_load(v, x + xr * ldx, lane, C);
        float mean, rstd;
        row_stats(v, lane, C, eps, mean, rstd);
        float4 g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
        LN_FOR(i, c) {
            const float4 d = load4<T>(dy + row * ldy + c);
            const float4 ww = *reinterpret_cast<const float4*>(w + c);
            v[i].x = (v[i].x - mean) * rstd; v[i].y = (v[i].y - mean) * rstd;
            v[i].z = (v[i].z - mean) * rstd; v[i].w = (v[i].w - mean) * rstd;
            pw[i].x += d.x * v[i].x; pw[i].y += d.y * v[i].y; pw[i].z += d.z * v[i].z; pw[i].w += d.w * v[i].w;
            pb[i].x += d.x; pb[i].y += d.y; pb[i].z += d.z; pb[i].w += d.w;
            g[i] = make_float4(d.x * ww.x, d.y * ww.y, d.z * ww.z, d.w * ww.w);
            s1 += g[i].x + g[i].y + g[i].z + g[i].w;
            s2 += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
        float* dr = dx + xr * lddx;
        LN_FOR(i, c) {
            float4 o = make_float4(rstd * (g[i].x - s1 - v[i].x * s2), rstd * (g[i].y - s1 - v[i].y * s2),
                                   rstd * (g[i].z - s1 - v[i].z * s2), rstd * (g[i].w - s1 - v[i].w * s2));
            if (accumulate) {
                const float4 p = *reinterpret_cast<const float4*>(dr + c);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            *reinterpret_cast<float4*>(dr + c) = o;
            if constexpr (CAST) {
                const uint32_t lo = pack_bf16x2(o.x, o.y), hi2 = pack_bf16x2(o.z, o.w);
                *reinterpret_cast<uint2*>(dxc + xr * ldc + c) = make_uint2(lo, hi2);
                pc[i].x += __uint_as_float(lo << 16); pc[i].y += __uint_as_float(lo & 0xFFFF0000u);
                pc[i].z += __uint_as_float(hi2 << 16); pc[i].w += __uint_as_float(hi2 & 0xFFFF0000u);
            }
        }
    }
    float* mine = lnb_red + wave * 2 * C;
    LN_FOR(i, c) {
        *reinterpret_cast<float4*>(mine + c) = pw[i];
        *reinterpret_cast<float4*>(mine + C + c) = pb[i];
    }
    __syncthreads();
    float* pr = partial + (long)blockIdx.x * NB * C;
    for (int c = threadIdx.x; c < 2 * C; c += 64 * LNB_WAVES) {
        float a = lnb_red[c];
#pragma unroll
        for (int w = 1; w < LNB_WAVES; ++w) a += lnb_red[w * 2 * C + c];
        pr[c] = a;
    }
    if constexpr (CAST) {                                     // third block through the same LDS (the 2 C layout keeps it at 48 KiB)
        __syncthreads();
        LN_FOR(i, c) *reinterpret_cast<float4*>(mine + c) = pc[i];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 64 * LNB_WAVES) {
            float a = lnb_red[c];
#pragma unroll
            for (int w = 1; w < LNB_WAVES; ++w) a += lnb_red[w * 2 * C + c];
            pr[2 * C + c] = a;
        }
    }
}

// second half of M324_AUX_N3: out[m][j] = bias[j] + sum_cb part[cb][m][j]; thread per output value, blocks added in order
__global__ __launch_bounds__(256) void n3_finish_kernel(const float* __restrict__ part, int ncb, int M, const float* __restrict__ bias3,
                                                        float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * 3) return;
    float s = bias3[i % 3];
    for (int cb = 0; cb < ncb; ++cb) s += part[(long)cb * M * 3 + i];
    out[i] = s;
}

// m324_attention_merge: the softmax over a key set that was attended in NP disjoint parts.  Part i left its normalised output O_i
// and the log2-domain log-sum-exp l_i of its scores (m324_attention's lse); the whole softmax is
//   O = sum_i 2^(l_i - m) O_i / sum_i 2^(l_i - m),  m = max_i l_i
// (each O_i is sum_k 2^(s_k - l_i) v_k over its own keys).  Token-major rows of H x 64 values, eight columns per lane.
template <typename T>
__global__ __launch_bounds__(256) void attention_merge_kernel(const T* __restrict__ O0, const float* __restrict__ l0, const T* __restrict__ O1,
                                                              const float* __restrict__ l1, const T* __restrict__ O2,
                                                              const float* __restrict__ l2, long ldp, T* __restrict__ O, long ldo, int B,
                                                              int H, int Lq) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // (row, 8-column chunk)
    const int cpr = H * 8;
    if (i >= (long)B * Lq * cpr) return;
    const long row = i / cpr;
    const int c = (int)(i - row * cpr), h = c >> 3;
    const long b = row / Lq, q = row - b * Lq;
    const long li = (b * H + h) * Lq + q;
    const float a0 = l0[li], a1 = l1[li], a2 = O2 ? l2[li] : -INFINITY;
    const float m = fmaxf(a0, fmaxf(a1, a2));
    float w0 = exp2f(a0 - m), w1 = exp2f(a1 - m), w2 = O2 ? exp2f(a2 - m) : 0.f;
    const float inv = 1.0f / (w0 + w1 + w2);
    w0 *= inv, w1 *= inv, w2 *= inv;
    float x[8], y[8];
    load8(O0 + row * ldp + c * 8, x);
    load8(O1 + row * ldp + c * 8, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = w0 * x[e] + w1 * y[e];
    if (O2) {
        load8(O2 + row * ldp + c * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaf(w2, y[e], x[e]);
    }
    store8(O + row * ldo + c * 8, x);
}

}  // namespace

#define DISPATCH_DTYPE(dtype, name, ...)                                  \
    if ((dtype) == M324_BF16) { using T = bf16_t; __VA_ARGS__; }          \
    else if ((dtype) == M324_F32) { using T = float; __VA_ARGS__; }       \
    else M324_FAIL(M324_ERR_UNSUPPORTED, name ": dtype %d", (int)(dtype))

extern "C" int m324_rowstats_finish(const float* part, int ncb, int M, float eps, float* rowstat, void* stream) {
    M324_REQUIRE(part && rowstat && ncb > 0 && M > 0, "m324_rowstats_finish: bad arguments (ncb=%d M=%d)", ncb, M);
    M324_REQUIRE(((uintptr_t)part % 8) == 0 && ((uintptr_t)rowstat % 8) == 0, "m324_rowstats_finish: tables must be 8-byte aligned");
    hipLaunchKernelGGL(rowstats_finish_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(part), ncb, M, eps, reinterpret_cast<float2*>(rowstat));
    M324_CHECK_LAUNCH("m324_rowstats_finish");
    return M324_OK;
}

extern "C" int m324_rowstats(const float* x, long ldx, int rows, int C, float eps, float* rowstat, void* copy, long ldcopy,
                             void* stream) {
    M324_REQUIRE(x && rowstat && rows > 0, "m324_rowstats: bad arguments");
    M324_REQUIRE(C % 4 == 0 && C > 0 && C <= 256 * LN_MAXV && ldx % 4 == 0 && ldx >= C, "m324_rowstats: C=%d ldx=%ld unsupported", C, ldx);
    M324_REQUIRE(((uintptr_t)rowstat % 8) == 0 && (!copy || (ldcopy % 4 == 0 && ldcopy >= C && ((uintptr_t)copy % 8) == 0)),
                 "m324_rowstats: misaligned outputs");
    hipLaunchKernelGGL(rowstats_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, eps,
                       reinterpret_cast<float2*>(rowstat), static_cast<bf16_t*>(copy), ldcopy);
    M324_CHECK_LAUNCH("m324_rowstats");
    return M324_OK;
}

extern "C" int m324_layernorm_in(const void* x, int x_dtype, long ldx, const float* w, const float* b, float eps, void* y, long ldy,
                                 int out_dtype, int rows, int C, int gin, int gout, int off, void* stream) {
    if (x_dtype == M324_F32) return m324_layernorm((const float*)x, ldx, w, b, eps, y, ldy, out_dtype, rows, C, gin, gout, off, stream);
    M324_REQUIRE(x && w && y, "m324_layernorm_in: null pointer");
    M324_REQUIRE(x_dtype == M324_BF16 && out_dtype == M324_BF16, "m324_layernorm_in: a bf16 input needs a bf16 output");
    M324_REQUIRE(rows > 0 && C % 4 == 0 && C > 0 && C <= 256 * LN_MAXV && ldx % 4 == 0 && ldy % 4 == 0,
                 "m324_layernorm_in: rows=%d C=%d ldx=%ld ldy=%ld unsupported", rows, C, ldx, ldy);
    if (m324::tunable(m324::TUN_LN_ROWS) == 1)
        hipLaunchKernelGGL((layernorm1_kernel<bf16_t, bf16_t>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, w, b, eps, (bf16_t*)y, ldy, rows, C, gin, gout, off);
    else
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), dim3(ceil_div(rows, 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, w, b, eps, (bf16_t*)y, ldy, rows, C, gin, gout, off);
    M324_CHECK_LAUNCH("m324_layernorm_in");
    return M324_OK;
}

extern "C" int m324_layernorm(const float* x, long ldx, const float* w, const float* b, float eps, void* y, long ldy,
                              int out_dtype, int rows, int C, int gin, int gout, int off, void* stream) {
    M324_REQUIRE(x && w && y, "m324_layernorm: null pointer");
    M324_REQUIRE(rows > 0, "m324_layernorm: rows=%d", rows);
    M324_REQUIRE(C % 4 == 0 && C > 0 && C <= 256 * LN_MAXV, "m324_layernorm: C=%d unsupported", C);
    M324_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "m324_layernorm: leading dims must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    if (m324::tunable(m324::TUN_LN_ROWS) == 1) {
        DISPATCH_DTYPE(out_dtype, "m324_layernorm",
                       hipLaunchKernelGGL(layernorm1_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, x, ldx, w, b, eps,
                                          (T*)y, ldy, rows, C, gin, gout, off));
    } else {
        DISPATCH_DTYPE(out_dtype, "m324_layernorm",
                       hipLaunchKernelGGL(layernorm_kernel<T>, dim3(ceil_div(rows, 8)), dim3(256), 0, s, x, ldx, w, b, eps,
                                          (T*)y, ldy, rows, C, gin, gout, off));
    }
    M324_CHECK_LAUNCH("m324_layernorm");
    return M324_OK;
}

extern "C" int m324_layernorm_pair(const float* x0, long ldx0, const float* w0, const float* b0, float eps0, void* y0, long ldy0, int rows0,
                                   int gin0, int gout0, int off0, const float* x1, long ldx1, const float* w1, const float* b1, float eps1,
                                   void* y1, long ldy1, int rows1, int gin1, int gout1, int off1, int C, int out_dtype, void* stream) {
    M324_REQUIRE(x0 && w0 && y0 && x1 && w1 && y1, "m324_layernorm_pair: null pointer");
    M324_REQUIRE(rows0 > 0 && rows1 > 0, "m324_layernorm_pair: rows=%d, %d", rows0, rows1);
    M324_REQUIRE(C % 4 == 0 && C > 0 && C <= 256 * LN_MAXV, "m324_layernorm_pair: C=%d unsupported", C);
    M324_REQUIRE(ldx0 % 4 == 0 && ldy0 % 4 == 0 && ldx1 % 4 == 0 && ldy1 % 4 == 0, "m324_layernorm_pair: leading dims must be multiples of 4");
    const LnProblem p0{x0, ldx0, w0, b0, eps0, y0, ldy0, rows0, gin0, gout0, off0}, p1{x1, ldx1, w1, b1, eps1, y1, ldy1, rows1, gin1, gout1, off1};
    const int blocks0 = ceil_div(rows0, 8);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(out_dtype, "m324_layernorm_pair",
                   hipLaunchKernelGGL(layernorm_pair_kernel<T>, dim3(blocks0 + ceil_div(rows1, 8)), dim3(256), 0, s, p0, p1, C, blocks0));
    M324_CHECK_LAUNCH("m324_layernorm_pair");
    return M324_OK;
}

extern "C" int m324_qkv_split(const void* q_src, long ldq, const void* k_src, long ldk, const void* v_src, long ldv,
                              const float* q_w, const float* k_w, float eps, float q_scale, void* Q, void* K, void* V, void* Qt,
                              void* Kt, void* Vt, int B, int L, int H, int dtype, void* stream) {
    M324_REQUIRE(B > 0 && L > 0 && H > 0, "m324_qkv_split: empty problem");
    M324_REQUIRE((!q_src || Q || Qt) && (!k_src || K || Kt) && (!v_src || V || Vt), "m324_qkv_split: missing output");
    const int al = dtype == M324_BF16 ? 8 : 4;          // 16-byte accesses
    M324_REQUIRE(ldq % al == 0 && ldk % al == 0 && ldv % al == 0, "m324_qkv_split: leading dims must be multiples of %d", al);
    const int Lp = (L + 63) / 64 * 64;
    dim3 grid(Lp / 64, H, B);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_qkv_split",
                   hipLaunchKernelGGL(qkv_split_kernel<T>, grid, dim3(256), 0, s, (const T*)q_src, ldq, (const T*)k_src, ldk,
                                      (const T*)v_src, ldv, q_w, k_w, eps, q_scale, (T*)Q, (T*)K, (T*)V, (T*)Qt, (T*)Kt, (T*)Vt,
                                      L, H, Lp));
    M324_CHECK_LAUNCH("m324_qkv_split");
    return M324_OK;
}

template <typename TIN>
static int patchify_launch(const TIN* video, int F, int Hin, int Win, int size, int patch, void* out, int Kp, int dtype, void* stream, const char* what) {
    M324_REQUIRE(video && out, "%s: null pointer", what);
    M324_REQUIRE(F > 0 && Hin > 0 && Win > 0 && size > 0 && patch > 0 && size % patch == 0, "%s: bad geometry", what);
    M324_REQUIRE(Kp >= 3 * patch * patch, "%s: Kp=%d too small", what, Kp);
    const long total = (long)F * size * size;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == M324_BF16)
        hipLaunchKernelGGL((patchify_kernel<bf16_t, TIN>), dim3(ceil_div(total, 256)), dim3(256), 0, s, video, F, Hin, Win, size, patch, (bf16_t*)out, Kp);
    else if (dtype == M324_F32)
        hipLaunchKernelGGL((patchify_kernel<float, TIN>), dim3(ceil_div(total, 256)), dim3(256), 0, s, video, F, Hin, Win, size, patch, (float*)out, Kp);
    else
        M324_FAIL(M324_ERR_UNSUPPORTED, "%s: dtype %d", what, dtype);
    M324_CHECK_LAUNCH(what);
    return M324_OK;
}

extern "C" int m324_patchify(const float* video, int F, int Hin, int Win, int size, int patch, void* out, int Kp, int dtype,
                             void* stream) {
    return patchify_launch<float>(video, F, Hin, Win, size, patch, out, Kp, dtype, stream, "m324_patchify");
}

extern "C" int m324_patchify_u8(const unsigned char* video, int F, int Hin, int Win, int size, int patch, void* out, int Kp, int dtype,
                                void* stream) {
    return patchify_launch<unsigned char>(video, F, Hin, Win, size, patch, out, Kp, dtype, stream, "m324_patchify_u8");
}

extern "C" int m324_point_encode(const float* xyz, int P, void* out, long ld, int dtype, void* stream) {
    M324_REQUIRE(xyz && out && P > 0 && ld >= 64, "m324_point_encode: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_point_encode",
                   hipLaunchKernelGGL(point_encode_kernel<T>, dim3(ceil_div((long)P * 64, 256)), dim3(256), 0, s, xyz, P,
                                      (T*)out, ld));
    M324_CHECK_LAUNCH("m324_point_encode");
    return M324_OK;
}

extern "C" int m324_point_concat(const float* normal, const float* rgb, int P, void* feat, int C, int Kp, int dtype,
                                 void* stream) {
    M324_REQUIRE(normal && rgb && feat && P > 0 && Kp >= C + 6, "m324_point_concat: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_point_concat",
                   hipLaunchKernelGGL(point_concat_kernel<T>, dim3(ceil_div((long)P * (Kp - C), 256)), dim3(256), 0, s,
                                      normal, rgb, P, (T*)feat, C, Kp));
    M324_CHECK_LAUNCH("m324_point_concat");
    return M324_OK;
}

extern "C" int m324_dino_cls_rows(const float* cls, const float* pos0, float* x, int F, int rows_per_frame, int C,
                                  void* stream) {
    M324_REQUIRE(cls && pos0 && x && F > 0 && C > 0, "m324_dino_cls_rows: bad arguments");
    hipLaunchKernelGGL(dino_cls_kernel, dim3(ceil_div((long)F * C, 256)), dim3(256), 0, (hipStream_t)stream, cls, pos0, x, F,
                       rows_per_frame, C);
    M324_CHECK_LAUNCH("m324_dino_cls_rows");
    return M324_OK;
}

extern "C" int m324_assemble_tokens(const float* dino_x, const float* dino_w, const float* dino_b, float eps_dino,
                                    const float* pos, const float* sp0, const float* spr, const float* mesh,
                                    const float* ln_w, float eps_in, float* out, int B, int T, int K, int P, int C,
                                    float drop_p, unsigned long long drop_seed, void* stream) {
    M324_REQUIRE(dino_x && dino_w && dino_b && pos && sp0 && spr && mesh && out, "m324_assemble_tokens: null pointer");
    M324_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "m324_assemble_tokens: drop_p=%g outside [0,1)", (double)drop_p);
    const unsigned drop_thr = (unsigned)(drop_p * 16777216.0f);
    const float drop_scale = 1.0f / (1.0f - drop_p);
    M324_REQUIRE(C % 4 == 0 && C <= 256 * LN_MAXV, "m324_assemble_tokens: C=%d unsupported", C);
    const long rows = (long)B * T * (4 + K + P);
    hipLaunchKernelGGL(assemble_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, dino_x, dino_w, dino_b,
                       eps_dino, pos, sp0, spr, mesh, ln_w, eps_in, out, B, T, K, P, C, drop_thr, drop_scale, drop_seed);
    M324_CHECK_LAUNCH("m324_assemble_tokens");
    return M324_OK;
}

extern "C" int m324_attention_merge(const void* O0, const float* lse0, const void* O1, const float* lse1, const void* O2, const float* lse2,
                                    long ldp, void* O, long ldo, int B, int H, int Lq, int dtype, void* stream) {
    M324_REQUIRE(O0 && lse0 && O1 && lse1 && O && (!O2 == !lse2), "m324_attention_merge: two or three (output, lse) parts");
    M324_REQUIRE(B > 0 && H > 0 && Lq > 0 && ldp >= (long)H * 64 && ldo >= (long)H * 64, "m324_attention_merge: bad sizes");
    const int esz = dtype == M324_BF16 ? 2 : 4;
    M324_REQUIRE((ldp * esz) % 16 == 0 && (ldo * esz) % 16 == 0 && ((uintptr_t)O0 % 16) == 0 && ((uintptr_t)O1 % 16) == 0 &&
                     ((uintptr_t)O2 % 16) == 0 && ((uintptr_t)O % 16) == 0,
                 "m324_attention_merge: rows must be 16-byte aligned");
    const long n = (long)B * Lq * H * 8;
    M324_REQUIRE((n + 255) / 256 < (1l << 31), "m324_attention_merge: grid too large");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_attention_merge",
                   hipLaunchKernelGGL(attention_merge_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const T*)O0, lse0,
                                      (const T*)O1, lse1, (const T*)O2, lse2, ldp, (T*)O, ldo, B, H, Lq));
    M324_CHECK_LAUNCH("m324_attention_merge");
    return M324_OK;
}

extern "C" int m324_n3_finish(const float* part, int ncb, int M, const float* bias3, float* out, void* stream) {
    M324_REQUIRE(part && bias3 && out && ncb > 0 && M > 0, "m324_n3_finish: bad arguments");
    hipLaunchKernelGGL(n3_finish_kernel, dim3(ceil_div((long)M * 3, 256)), dim3(256), 0, (hipStream_t)stream, part, ncb, M, bias3, out);
    M324_CHECK_LAUNCH("m324_n3_finish");
    return M324_OK;
}

extern "C" int m324_linear_n3(const void* A, long lda, const float* W, const float* bias, float* out, int M, int K, int dtype,
                              void* stream) {
    M324_REQUIRE(A && W && bias && out && M > 0, "m324_linear_n3: bad arguments");
    M324_REQUIRE(K % 4 == 0 && lda % 4 == 0, "m324_linear_n3: K and lda must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_linear_n3",
                   hipLaunchKernelGGL(linear_n3_kernel<T>, dim3(ceil_div(M, 4)), dim3(256), 0, s, (const T*)A, lda, W, bias,
                                      out, M, K));
    M324_CHECK_LAUNCH("m324_linear_n3");
    return M324_OK;
}

extern "C" int m324_mse(const float* pred, const float* target, long n, float weight, float* partial, float* out,
                        void* stream) {
    M324_REQUIRE(pred && target && partial && out && n > 0, "m324_mse: bad arguments");
    const int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, s, pred, target, n, partial);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, s, partial, nb, n, weight, out);
    M324_CHECK_LAUNCH("m324_mse");
    return M324_OK;
}

extern "C" int m324_smooth_trajectories(const float* trajs, float* tmp, float* out, int B, int T, int N, float threshold,
                                        float sigma, void* stream) {
    M324_REQUIRE(trajs && tmp && out && B > 0 && T > 0 && N > 0, "m324_smooth_trajectories: bad arguments");
    M324_REQUIRE(trajs != out && tmp != out && trajs != tmp, "m324_smooth_trajectories: buffers must not alias");
    hipStream_t s = (hipStream_t)stream;
    const bool gauss = sigma > 0.f;
    float* stage1 = gauss ? tmp : out;
    hipLaunchKernelGGL(smooth_threshold_kernel, dim3(ceil_div((long)B * N, 256)), dim3(256), 0, s, trajs, stage1, B, T, N,
                       threshold);
    if (gauss) {
        const int radius = (int)(4.0f * sigma + 0.5f);
        hipLaunchKernelGGL(smooth_gauss_kernel, dim3(ceil_div((long)B * T * N, 256)), dim3(256), 0, s, stage1, out, B, T, N,
                           sigma, radius);
    }
    M324_CHECK_LAUNCH("m324_smooth_trajectories");
    return M324_OK;
}

extern "C" int m324_smooth_savgol(const float* trajs, float* out, int B, int T, int N, const double* coef, int window,
                                  void* stream) {
    M324_REQUIRE(trajs && out && coef && B > 0 && T > 0 && N > 0 && window >= 1 && (window & 1) && trajs != out,
                 "m324_smooth_savgol: bad arguments (odd window, distinct buffers)");
    M324_REQUIRE(T >= window, "m324_smooth_savgol: T=%d is shorter than the window %d (the reference leaves such clips unfiltered)", T, window);
    hipLaunchKernelGGL(smooth_fir_kernel, dim3(ceil_div((long)B * T * N, 256)), dim3(256), 0, (hipStream_t)stream, trajs, out, B, T,
                       N, coef, window);
    M324_CHECK_LAUNCH("m324_smooth_savgol");
    return M324_OK;
}

extern "C" int m324_smooth_oneeuro(const float* trajs, float* out, int B, int T, int N, float mincutoff, float beta, float dcutoff,
                                   void* stream) {
    M324_REQUIRE(trajs && out && B > 0 && T > 0 && N > 0 && trajs != out, "m324_smooth_oneeuro: bad arguments");
    hipLaunchKernelGGL(smooth_oneeuro_kernel, dim3(ceil_div((long)B * N * 3, 256)), dim3(256), 0, (hipStream_t)stream, trajs, out, B,
                       T, N, (double)mincutoff, (double)beta, (double)dcutoff);
    M324_CHECK_LAUNCH("m324_smooth_oneeuro");
    return M324_OK;
}

extern "C" int m324_nearest_point(const float* query, int n_query, const float* ref, int n_ref, int* index, void* stream) {
    M324_REQUIRE(query && ref && index && n_query > 0 && n_ref > 0, "m324_nearest_point: bad arguments");
    hipLaunchKernelGGL(nearest_point_kernel, dim3(ceil_div(n_query, 256)), dim3(256), 0, (hipStream_t)stream, query, n_query, ref,
                       n_ref, index);
    M324_CHECK_LAUNCH("m324_nearest_point");
    return M324_OK;
}

extern "C" int m324_transpose(const void* in, long ld_in, void* out, long ld_out, int rows, int cols, int rows_pad, int dtype,
                              void* stream) {
    M324_REQUIRE(in && out && rows > 0 && cols > 0 && rows_pad >= rows && ld_in >= cols && ld_out >= rows_pad,
                 "m324_transpose: bad arguments");
    dim3 grid(ceil_div(cols, 64), ceil_div(rows_pad, 64));
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, "m324_transpose",
                   hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, s, (const T*)in, ld_in, (T*)out, ld_out, rows, cols,
                                      rows_pad));
    M324_CHECK_LAUNCH("m324_transpose");
    return M324_OK;
}

extern "C" int m324_colsum(const void* x, long ld, float* out, int rows, int cols, int dtype, int accumulate, float* scratch,
                           int scratch_rows, void* stream) {
    M324_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "m324_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // four columns per lane: 8-byte (bf16) / 16-byte (fp32) accesses need whole 4-column groups and 16-byte aligned rows
    const bool v4 = cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && (dtype == M324_BF16 || dtype == M324_F32);
    if (scratch && scratch_rows > 1 && rows >= 256) {    // two-stage: row chunks in parallel, then a short deterministic sum
        const int R = scratch_rows < 256 ? scratch_rows : 256;
        if (v4 && ((uintptr_t)scratch % 16) == 0) {
            DISPATCH_DTYPE(dtype, "m324_colsum",
                           hipLaunchKernelGGL(colsum_chunk4_kernel<T>, dim3(ceil_div(cols, 256), R), dim3(256), 0, s, (const T*)x, ld,
                                              scratch, rows, cols));
            const int nb = ceil_div(cols, 1024) < 8192 ? ceil_div(cols, 1024) : 8192;
            hipLaunchKernelGGL(colsum_wide4_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)scratch, (long)cols, out, R, (long)cols,
                               accumulate);
            M324_CHECK_LAUNCH("m324_colsum");
            return M324_OK;
        }
        DISPATCH_DTYPE(dtype, "m324_colsum",
                       hipLaunchKernelGGL(colsum_chunk_kernel<T>, dim3(ceil_div(cols, 64), R), dim3(256), 0, s, (const T*)x, ld,
                                          scratch, rows, cols));
        const int nb = (cols + 255) / 256 < 8192 ? (cols + 255) / 256 : 8192;
        hipLaunchKernelGGL(colsum_wide_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)scratch, (long)cols, out, R,
                           (long)cols, accumulate);
        M324_CHECK_LAUNCH("m324_colsum");
        return M324_OK;
    }
    if (rows <= 64 && v4) {
        const long q = ((long)cols + 1023) / 1024;
        DISPATCH_DTYPE(dtype, "m324_colsum",
                       hipLaunchKernelGGL(colsum_wide4_kernel<T>, dim3((int)(q < 8192 ? q : 8192)), dim3(256), 0, s, (const T*)x, ld, out, rows,
                                          (long)cols, accumulate));
    } else if (rows <= 64) {
        const int nb = (int)(((long)cols + 255) / 256 < 8192 ? ((long)cols + 255) / 256 : 8192);
        DISPATCH_DTYPE(dtype, "m324_colsum",
                       hipLaunchKernelGGL(colsum_wide_kernel<T>, dim3(nb), dim3(256), 0, s, (const T*)x, ld, out, rows, (long)cols,
                                          accumulate));
    } else {
        DISPATCH_DTYPE(dtype, "m324_colsum",
                       hipLaunchKernelGGL(colsum_kernel<T>, dim3(ceil_div(cols, 64)), dim3(256), 0, s, (const T*)x, ld, out, rows,
                                          cols, accumulate));
    }
    M324_CHECK_LAUNCH("m324_colsum");
    return M324_OK;
}

extern "C" int m324_colsum_multi(const m324_colsum_item* items, int n, void* stream) {
    M324_REQUIRE(items && n > 0, "m324_colsum_multi: no items");
    M324_REQUIRE(!items[0].chain, "m324_colsum_multi: the first item cannot continue a chain");
    for (int i = 0; i < n; ++i) {
        const m324_colsum_item& it = items[i];
        M324_REQUIRE(it.dst && it.src && it.rows > 0 && it.cols > 0 && it.ld >= it.cols, "m324_colsum_multi: item %d: bad arguments", i);
        M324_REQUIRE(!it.chain || (it.dst == items[i - 1].dst && it.cols == items[i - 1].cols), "m324_colsum_multi: item %d continues another destination", i);
    }
    hipStream_t s = (hipStream_t)stream;
    int i0 = 0;
    while (i0 < n) {
        // as many whole chains as fit the table; a chain longer than the table is cut: its continuation accumulates into the destination
        CsmTable t;
        int cnt = 0, blocks = 0, i = i0;
        bool cut = false;
        while (i < n && cnt < CSM_MAX) {
            int j = i + 1;
            while (j < n && items[j].chain) ++j;                       // chain [i, j)
            if (cnt + (j - i) > CSM_MAX) {
                if (cnt > 0) break;                                    // next launch starts with this chain
                j = i + CSM_MAX, cut = true;                           // one chain longer than the table
            }
            bool tall = false, vec = true;
            for (int k = i; k < j; ++k) {
                tall = tall || items[k].rows > 64;
                vec = vec && (items[k].ld & 3) == 0 && (((uintptr_t)items[k].src) & 15) == 0;
            }
            vec = vec && (items[i].cols & 3) == 0 && (((uintptr_t)items[i].dst) & 15) == 0;
            const int nb = tall ? ceil_div(items[i].cols, 64) : vec ? ceil_div(items[i].cols, 4 * CSM_THREADS) : ceil_div(items[i].cols, CSM_THREADS);
            for (int k = i; k < j; ++k) {
                t.it[cnt] = items[k];
                if (k == i && i == i0 && i0 > 0 && items[i].chain) t.it[cnt].chain = 0, t.it[cnt].accumulate = 1;   // continuation of a cut chain
                t.blk0[cnt] = blocks;
                ++cnt;
            }
            blocks += nb;
            i = j;
            if (cut) break;
        }
        for (int k = cnt; k <= CSM_MAX; ++k) t.blk0[k] = blocks;
        t.n = cnt;
        hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(CSM_THREADS), 0, s, t);
        M324_CHECK_LAUNCH("m324_colsum_multi");
        i0 = i;
    }
    return M324_OK;
}

extern "C" int m324_gelu(const void* z, void* h, long n, int dtype, void* stream) {
    M324_REQUIRE(z && h && n > 0, "m324_gelu: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n % 8 == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)h % 16) == 0) {
        const int nb8 = (int)((n / 8 + 255) / 256 < 8192 ? (n / 8 + 255) / 256 : 8192);
        DISPATCH_DTYPE(dtype, "m324_gelu", hipLaunchKernelGGL(gelu_fwd8_kernel<T>, dim3(nb8), dim3(256), 0, s, (const T*)z, (T*)h, n / 8));
        M324_CHECK_LAUNCH("m324_gelu");
        return M324_OK;
    }
    const int nb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DISPATCH_DTYPE(dtype, "m324_gelu", hipLaunchKernelGGL(gelu_fwd_kernel<T>, dim3(nb), dim3(256), 0, s, (const T*)z, (T*)h, n));
    M324_CHECK_LAUNCH("m324_gelu");
    return M324_OK;
}

extern "C" int m324_gelu_bwd(const void* z, const void* dh, void* dz, long n, int dtype, void* stream) {
    M324_REQUIRE(z && dh && dz && n > 0, "m324_gelu_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n % 8 == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)dh % 16) == 0 && ((uintptr_t)dz % 16) == 0) {
        const int nb8 = (int)((n / 8 + 255) / 256 < 8192 ? (n / 8 + 255) / 256 : 8192);
        DISPATCH_DTYPE(dtype, "m324_gelu_bwd",
                       hipLaunchKernelGGL(gelu_bwd8_kernel<T>, dim3(nb8), dim3(256), 0, s, (const T*)z, (const T*)dh, (T*)dz, n / 8));
        M324_CHECK_LAUNCH("m324_gelu_bwd");
        return M324_OK;
    }
    const int nb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DISPATCH_DTYPE(dtype, "m324_gelu_bwd",
                   hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(nb), dim3(256), 0, s, (const T*)z, (const T*)dh, (T*)dz, n));
    M324_CHECK_LAUNCH("m324_gelu_bwd");
    return M324_OK;
}

extern "C" int m324_layernorm_bwd(const float* x, long ldx, const float* w, float eps, const void* dy, long ldy, int dy_dtype,
                                  float* dx, long lddx, int accumulate, float* partial, int n_partial, int rows, int C, int gin,
                                  int gout, int off, void* stream) {
    M324_REQUIRE(x && w && dy && dx && partial, "m324_layernorm_bwd: null pointer");
    M324_REQUIRE(rows > 0 && C % 4 == 0 && C > 0 && C <= 256 * LN_MAXV, "m324_layernorm_bwd: rows=%d C=%d unsupported", rows, C);
    M324_REQUIRE(n_partial > 0 && n_partial <= 4096, "m324_layernorm_bwd: n_partial = %d (1 .. 4096 workgroups)", n_partial);
    M324_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && lddx % 4 == 0, "m324_layernorm_bwd: leading dims must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dy_dtype, "m324_layernorm_bwd",
                   hipLaunchKernelGGL((layernorm_bwd_kernel<T, false>), dim3(n_partial), dim3(64 * LNB_WAVES), (size_t)LNB_WAVES * 2 * C * 4, s, x,
                                      ldx, w, eps, (const T*)dy, ldy, dx, lddx, accumulate, partial, rows, C, gin, gout, off,
                                      (bf16_t*)nullptr, 0l));
    M324_CHECK_LAUNCH("m324_layernorm_bwd");
    return M324_OK;
}

extern "C" int m324_layernorm_bwd_cast(const float* x, long ldx, const float* w, float eps, const void* dy, long ldy, int dy_dtype,
                                       float* dx, long lddx, int accumulate, float* partial, int n_partial, int rows, int C, int gin,
                                       int gout, int off, void* dx_bf16, long ldc, void* stream) {
    M324_REQUIRE(x && w && dy && dx && partial && dx_bf16, "m324_layernorm_bwd_cast: null pointer");
    M324_REQUIRE(rows > 0 && C % 4 == 0 && C > 0 && C <= 256 * LN_MAXV, "m324_layernorm_bwd_cast: rows=%d C=%d unsupported", rows, C);
    M324_REQUIRE(n_partial > 0 && n_partial <= 4096, "m324_layernorm_bwd_cast: n_partial = %d (1 .. 4096 workgroups)", n_partial);
    M324_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && lddx % 4 == 0 && ldc % 4 == 0, "m324_layernorm_bwd_cast: leading dims must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dy_dtype, "m324_layernorm_bwd_cast",
                   hipLaunchKernelGGL((layernorm_bwd_kernel<T, true>), dim3(n_partial), dim3(64 * LNB_WAVES), (size_t)LNB_WAVES * 2 * C * 4, s, x,
                                      ldx, w, eps, (const T*)dy, ldy, dx, lddx, accumulate, partial, rows, C, gin, gout, off,
                                      (bf16_t*)dx_bf16, ldc));
    M324_CHECK_LAUNCH("m324_layernorm_bwd_cast");
    return M324_OK;
}

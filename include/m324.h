/*
 * libm324 -- C ABI of the MI355X-native (gfx950) kernels behind Motion324's per-frame
 * motion-prediction hot path (Motion_Latent_Model.forward).
 *
 * The reference has no FFI of its own on this path: its arithmetic is issued through torch.nn,
 * xformers' flash-attention op and the torch.hub DINOv2 module.  Each entry point below states
 * the reference call site it replaces (paths relative to the reference repo).  A maintainer binds
 * them with ctypes; motion324_amd/lib.py is that binding.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (PyTorch
 *     allocations in practice); the library allocates nothing persistent.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it.
 *   - return 0 on success, negative m324_status on failure; m324_last_error() gives the text of
 *     the calling thread's last failure.  Nothing throws or aborts across the boundary.
 *   - dtype codes: M324_F32 = fp32 "parity" arithmetic (f32 MFMA, exact), M324_BF16 = bf16 operands
 *     with fp32 accumulation ("speed" mode, what the reference runs under torch.autocast(bf16)).
 *   - matrices are row-major; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef M324_H
#define M324_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { M324_OK = 0, M324_ERR_INVALID = -1, M324_ERR_HIP = -2, M324_ERR_UNSUPPORTED = -3 } m324_status;
typedef enum { M324_F32 = 0, M324_BF16 = 1 } m324_dtype;
typedef enum { M324_ACT_NONE = 0, M324_ACT_GELU = 1 } m324_act;
/* aux_mode of m324_gemm (aux has the output dtype and layout [M, ldaux], no row remap):
 *   M324_AUX_STORE_PREACT  : aux[m,n] = acc + bias, the value the activation is applied to -- one launch gives the
 *                            MLP both gelu(z) (C) and z (aux), which the backward needs (autograd of transformer.py:73-78);
 *   M324_AUX_MUL_GELU_GRAD : the result is multiplied by gelu'(aux[m,n]) = Phi(z) + z phi(z) before it is stored -- the
 *                            dgrad GEMM of the MLP's second Linear then delivers d(pre-activation) directly.
 *   M324_AUX_STORE_GELU_GRAD / M324_AUX_MUL (ABI 22): the same pair with the work moved to where erf is evaluated anyway --
 *                            aux[m,n] = gelu'(acc + bias) next to C = gelu(acc + bias), and "the result is multiplied by aux[m,n]":
 *                            the dgrad epilogue multiplies instead of evaluating erf and exp a second time.                    */
typedef enum { M324_AUX_NONE = 0, M324_AUX_STORE_PREACT = 1, M324_AUX_MUL_GELU_GRAD = 2, M324_AUX_QKV_HEADS = 3,
               M324_AUX_QKV_HEADS_VT = 4, M324_AUX_N3 = 5, M324_AUX_STORE_GELU_GRAD = 6, M324_AUX_MUL = 7 } m324_aux_mode;
/*   M324_AUX_N3 (inference, bf16): the regression head Linear -> GELU -> Linear(N -> 3) (Pcd_motion.py:336-341) without its
 *   [M, N] intermediate: h = gelu(A W^T + bias) is contracted with the [3, N] fp32 weight passed in qkv_qw inside the
 *   epilogue and only partial sums leave, aux = float part[N / 64][M][3] (C may be NULL); m324_n3_finish adds the N / 64
 *   column blocks in a fixed order and the last layer's bias.  N % 256 == 0, K % 64 == 0, K >= 128.                       */

/* ABI version of this header (bumped on any signature change). */
int m324_abi_version(void);
/* Copies the calling thread's last error text into buf (NUL-terminated); returns its length. */
int m324_last_error(char* buf, int n);
/* Fills name with the device's gcnArchName ("gfx950..."), returns CU count or negative status. */
int m324_device_info(char* name, int n);
/* Lab / test hook (not used by the product path): the kernel choosers' A/B switches -- "M324_GEMM" (forced schedule
 * number, 0 = automatic), "M324_GEMM_TN", "M324_XCD", "M324_ATTN_NW", "M324_ATTN_FLAT", "M324_ATTN_OCC", "M324_ATTN_NQ2",
 * "M324_ATTN_BWD_NW", "M324_ATTN_EXP", "M324_LN_ROWS", "M324_GEMM_PERSIST" -- are read from the environment ONCE, when the library is loaded; this call
 * overrides one of them afterwards (value INT_MIN restores the default).  No launch path calls getenv(). */
int m324_set_tunable(const char* name, int value);

/* ------------------------------------------------------------------------------------------
 * m324_gemm: C = epilogue(A[M,K] . W[N,K]^T)  -- every nn.Linear / the patch Conv2d on the path.
 *   replaces: F.linear in model/transformer.py:112-116,182-183,73-78 (to_q/to_k/to_v/to_qkv/fc/MLP),
 *             model/Pcd_motion.py:175,284,338-340 (point_embed.mlp, point_normal_rgb_proj, head),
 *             DINOv2 patch_embed.proj / qkv / proj / fc1 / fc2 (model/image_encoder/dino/model_dino.py:160-170,
 *             174-231,338-354).
 *   epilogue, in this order:  v = acc (+ bias[n]) ; v = gelu_erf(v) if act ; v *= gamma[n] ;
 *                             v += residual[(m % res_rows) * ldr + n]  (fp32 -- except when `residual` IS `C` and out_dtype is
 *                             bf16: then it is the bf16 stream being updated in place) ; store as out_dtype at
 *                             C[out_row(m) * ldc + n],  out_row(m) = (m / row_gin) * row_gout + m % row_gin + row_off.
 *   constraints: K % 64 == 0 (bf16) / K % 32 == 0 (f32) -- pad K with zeros; A, W rows 16-byte aligned.
 *   batch > 1 runs independent problems in one launch (used for split-K weight gradients: the token dimension is cut
 *   into slices, each slice writes its own partial dW, m324_colsum adds them -- deterministic, no atomics).
 *   in_dtype selects the MFMA: bf16 -> v_mfma_f32_32x32x16_bf16, f32 -> v_mfma_f32_32x32x2_f32.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* A;  long lda;
    const void* W;  long ldw;
    void* C;        long ldc;
    int M, N, K;
    int in_dtype, out_dtype;
    const float* bias;
    int act;
    const float* gamma;
    const float* residual; long ldr; int res_rows;   /* res_rows <= 0 -> M */
    int row_gin, row_gout, row_off;                  /* row_gin <= 0 -> identity */
    int batch;                                       /* <= 1: single GEMM; else `batch` independent GEMMs ...   */
    long strideA, strideW, strideC;                  /* ... whose A / W / C start strideX elements apart          */
    void* aux; long ldaux; int aux_mode;             /* training: second operand of the epilogue, see M324_AUX_*  */
    /* M324_AUX_QKV_HEADS (inference, bf16): the [M, 3*H*64] result of a fused q|k|v projection is not stored token-major
     * in C (C may be NULL) but split into head-major qkv_q / qkv_k / qkv_v [B, H, L, 64] (row m = b * qkv_L + l), with
     * the per-head RMSNorm of q and k (weights qkv_qw / qkv_kw [64], NULL = none, eps qkv_eps) and q * qkv_qscale
     * applied on the fp32 accumulators -- what m324_qkv_split does in a second pass (transformer.py:36-42,200-207).   
     * M324_AUX_QKV_HEADS_VT: the same, but qkv_v receives the TRANSPOSED, key-permuted Vt [B, H, 64, qkv_L] that
     * m324_attention reads by default (m324_qkv_split's Vt; needs qkv_L % 64 == 0, so there is no padding).
     * Cross-attention projections (transformer.py:112-132): N = 2 H 64 with qkv_q NULL is the k|v projection, N = H 64
     * with qkv_k and qkv_v NULL the q projection.                                                                      */
    void* qkv_q; void* qkv_k; void* qkv_v;
    const float* qkv_qw; const float* qkv_kw;
    float qkv_eps, qkv_qscale;
    int qkv_L, qkv_H;
    /* LayerNorm fold (bf16 inference): the LayerNorm in front of a Linear (transformer.py:365-376,400-423; DINOv2 norm1 /
     * norm2; Pcd_motion.py:336-338) without its own pass over the residual stream.
     *   consumer: A holds the RAW stream x (bf16) and W the weight with the LayerNorm scale folded in, W'[n,k] = w_ln[k] W[n,k];
     *     ln_rowstat[m] = (rstd_m, -rstd_m mean_m), ln_colsum[n] = sum_k W'[n,k] (of the rounded bf16 values), and `bias` is
     *     b[n] + sum_k b_ln[k] W[n,k].  The epilogue starts from  rstd_m acc - rstd_m mean_m colsum[n] + bias[n]
     *     = Linear(LayerNorm(x_m)) and continues as usual (activation, q|k|v heads, N3 ...).
     *   producer: ln_stats_out[cb][m] = (sum, sum of squared deviations from the block mean) over the 64 columns of block cb
     *     of the values row m of C receives (fp32, before they are rounded to out_dtype); m324_rowstats_finish merges the
     *     N / 64 blocks of a row (Chan's update: no E[x^2] - mean^2 cancellation) into ln_rowstat.  ln_copy_out (fp32 C only):
     *     the same values once more as bf16 [M, ln_ldcopy] -- the consumer's A operand.
     *   Needs M > 64, N % 64 == 0, K >= 128, no batch.  Built combinations: a consumer (ln_rowstat) has no residual, gamma
     *   or row map, and a bf16 output behind GELU; a producer (ln_stats_out) is a residual update without activation, row
     *   map or aux mode, with ln_copy_out exactly when C is fp32; one GEMM is never both. */
    const float* ln_rowstat; const float* ln_colsum;
    float* ln_stats_out;
    void* ln_copy_out; long ln_ldcopy;
    /*   consumer, unmerged table (ln_ncb > 0): ln_rowstat is not the merged table but the producer's ln_stats_out itself,
     *     [ln_ncb][M] (sum, M2) with ln_ncb = K / 64 even and <= 16, and the consumer merges the blocks of its tiles' rows (the arithmetic
     *     of m324_rowstats_finish, eps = ln_eps) while its first operand tiles are in flight -- no launch between the two GEMMs. */
    int ln_ncb; float ln_eps;
} m324_gemm_args;
int m324_gemm(const m324_gemm_args* a, void* stream);
/* Two independent GEMMs in ONE launch (horizontal fusion of small latency-bound problems).  Built for the pair the hot path has:
 * two bf16 projections with the head-major q|k|v epilogue that m324_gemm would run on the 128 x 128 chunk ring -- the decoder's
 * q projection of the mesh points and k|v projection of the latent tokens (transformer.py:112-132 under Pcd_motion.py:556-560).
 * Any other pair returns M324_ERR_UNSUPPORTED without launching anything: issue two m324_gemm calls. */
int m324_gemm_pair(const m324_gemm_args* a, const m324_gemm_args* b, void* stream);
/* Host-only: writes the kernel symbol (as rocprofv3 prints the template) and its grid in threads that m324_gemm would
 * launch for `a` into buf; returns the schedule number.  bench.py labels its per-launch HIP-event rows with it so that
 * they can be matched against the committed rocprofv3 summaries (profiles/). */
int m324_gemm_plan(const m324_gemm_args* a, char* buf, int n);

/* LayerNorm fold, between producer and consumer: rowstat[m] = (rstd, -rstd mean) of row m from the ncb per-block
 * (sum, M2) pairs part[cb][m] a producer GEMM left (blocks of 64 columns, C = 64 ncb);  eps as in nn.LayerNorm. */
int m324_rowstats_finish(const float* part, int ncb, int M, float eps, float* rowstat, void* stream);
/* The same table straight from an fp32 stream x [rows, C] (the head of a chain: token assembly / patch embedding
 * output), plus the bf16 copy [rows, ldcopy] a folded consumer reads (copy may be NULL).  C % 4 == 0, C <= 1024. */
int m324_rowstats(const float* x, long ldx, int rows, int C, float eps, float* rowstat, void* copy, long ldcopy, void* stream);

/* out[m][j] = bias3[j] + sum over the ncb column blocks of part[cb][m][j]  (the second half of M324_AUX_N3). */
int m324_n3_finish(const float* part, int ncb, int M, const float* bias3, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_gemm_tn: C[s][n][j] (fp32) = sum over the tokens m of slice s of X[m, n] * Y[m, j]  -- the weight gradient
 *   dW = dY^T . A of every nn.Linear (autograd of transformer.py:73-78,112-116,182-183 under train.py:166), computed
 *   straight from the token-major bf16 activations (no transposed copies).  X [M, ldx] (N columns), Y [M, ldy] (Kc columns),
 *   bf16; the M tokens are cut into `slices` ranges of whole 64-row tiles, slice s writes its partial at C + s * strideC
 *   ([N, ldc]); the caller sums the partials (m324_colsum over [slices, N * ldc]) -- deterministic, no atomics.
 *   N % 8 == 0, Kc % 8 == 0, ldx / ldy multiples of 8, 16-byte aligned bases.
 * ------------------------------------------------------------------------------------------ */
int m324_gemm_tn(const void* X, long ldx, const void* Y, long ldy, float* C, long ldc, int M, int N, int Kc,
                 int slices, long strideC, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_layernorm: y = (x - mean) * rsqrt(var + eps) * w (+ b) over the last dim of fp32 x.
 *   replaces: nn.LayerNorm at model/transformer.py:345-346,357,400,411 (bias=False, eps 1e-5),
 *             model/Pcd_motion.py:326,337 and DINOv2 norm1/norm2/norm (eps 1e-6, with bias).
 *   input row r is read from x[in_row(r) * ldx], in_row(r) = (r / gin) * gout + r % gin + off
 *   (gin <= 0 -> identity): lets the decoder normalise the 64 latent tokens of every frame in place
 *   (Pcd_motion.py:520, transformer.py:368-369) without a gather copy.
 *   C % 4 == 0, C <= 1024.
 * ------------------------------------------------------------------------------------------ */
int m324_layernorm(const float* x, long ldx, const float* w, const float* b, float eps,
                   void* y, long ldy, int out_dtype, int rows, int C,
                   int gin, int gout, int off, void* stream);
/* Two of them (same C, same out_dtype) in one launch: problem 0 without row map, problem 1 with (gin1, gout1, off1) -- the decoder's
 * norm_q over the mesh points and norm_kv over the gathered latent tokens (transformer.py:365-369). */
int m324_layernorm_pair(const float* x0, long ldx0, const float* w0, const float* b0, float eps0, void* y0, long ldy0, int rows0,
                        int gin0, int gout0, int off0, const float* x1, long ldx1, const float* w1, const float* b1, float eps1,
                        void* y1, long ldy1, int rows1, int gin1, int gout1, int off1, int C, int out_dtype, void* stream);
/* The same with the input's dtype given: x_dtype = M324_BF16 (bf16 output only) reads a bf16 residual stream -- the
 * decoder's in bf16 inference, where x is two additions deep and its LayerNorm output is rounded to bf16 anyway. */
int m324_layernorm_in(const void* x, int x_dtype, long ldx, const float* w, const float* b, float eps,
                      void* y, long ldy, int out_dtype, int rows, int C,
                      int gin, int gout, int off, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_qkv_split: head-major operands for m324_attention from token-major projections.
 *   replaces: rearrange "b l (nh dh) -> b l nh dh" + RMSNorm(q), RMSNorm(k)
 *             (model/transformer.py:128-132,200-207; RMSNorm :36-42, eps 1e-5, fp32 inside).
 *   q_src/k_src/v_src: [B*L, *] rows of `dtype` with leading dims ldq/ldk/ldv (any may be NULL);
 *   row b*L + l, columns h*64 .. h*64+63.  q_w / k_w: RMSNorm weights [64] or NULL (DINO: no qk-norm).
 *   Outputs (same dtype): Q[B,H,L,64], K[B,H,L,64], Vt[B,H,64,Lp] with Lp = round_up(L, 64);
 *   Vt columns L..Lp-1 are written as zeros.  head_dim is fixed at 64 (config d_head).
 *   Vt key order: inside every aligned group of 16 keys the columns are stored as keys 0-3, 8-11, 4-7, 12-15
 *   (the order the attention MFMA contracts them in); m324_attention expects exactly this layout.
 *   q_scale multiplies the (normalised) q before it is rounded to `dtype`: pass softmax_scale * log2(e)
 *   and call m324_attention with q_prescaled = 1 (saves one multiply per score in the kernel), or 1.0f.
 * ------------------------------------------------------------------------------------------ */
int m324_qkv_split(const void* q_src, long ldq, const void* k_src, long ldk, const void* v_src, long ldv,
                   const float* q_w, const float* k_w, float eps, float q_scale,
                   void* Q, void* K, void* V, void* Qt, void* Kt, void* Vt,
                   int B, int L, int H, int dtype, void* stream);
/*   Each source may be emitted row-major (Q, K, V: [B,H,L,64]) and/or transposed (Qt, Kt, Vt: [B,H,64,Lp], the
 *   permuted, zero-padded layout described above); NULL outputs are skipped.  Inference uses Q, K, Vt; the attention
 *   backward also uses V, Qt, Kt and the same kernel on dO. */

/* ------------------------------------------------------------------------------------------
 * m324_attention: O = softmax(Q K^T * scale) V, flash-style (no L x L matrix in HBM).
 *   replaces: xformers.ops.memory_efficient_attention(q,k,v, op=flash) at model/transformer.py:134-139,
 *             209-214 and DINOv2 self-attention (model_dino.py:200-222); scale = 64^-0.5.
 *   Q[Bq,H,Lq,64] (q_bstride elements between batches; 0 = one query set shared by every batch, as the
 *   decoder does with the mesh points, Pcd_motion.py:534-560), K[B,H,Lk,64], Vt[B,H,64,Lkp]
 *   (Lkp = round_up(Lk,64), zero padded).  O[B, Lq, H*64] token-major (ldo = row stride), same dtype.
 *   q_prescaled is a flag word: M324_ATTN_Q_PRESCALED (1): Q already holds q * scale * log2(e) (see m324_qkv_split) and
 *   `scale` is ignored; M324_ATTN_V_ROWMAJOR (2, bf16 only): the `Vt` argument is row-major V[B,H,Lk,64] (not transposed,
 *   not padded) and the kernel transposes its fragments in the LDS read; M324_ATTN_SCORES_BOUNDED (4): the caller vouches
 *   that every log2-domain score |q . k * scale * log2(e)| is <= 64 (per-head RMSNorm bounds it by 64 max|w_q| max|w_k|
 *   scale log2(e), transformer.py:36-42): kernels that have such a form (the long-sequence kernel attention_pwg.hip) then run
 *   the softmax without a reference maximum -- exp2, sum and bf16 pack only; the others ignore the flag.  Results are the
 *   same softmax (2^s / sum 2^s needs no shift inside fp32 / bf16 range); scores beyond the vouched bound overflow.
 * ------------------------------------------------------------------------------------------ */
enum { M324_ATTN_Q_PRESCALED = 1, M324_ATTN_V_ROWMAJOR = 2, M324_ATTN_SCORES_BOUNDED = 4 };
int m324_attention(const void* Q, long q_bstride, const void* K, const void* Vt, void* O, long ldo,
                   int B, int H, int Lq, int Lk, float scale, int q_prescaled, float* lse, int dtype, void* stream);
/* Host-only twin of m324_gemm_plan for m324_attention (flags = the q_prescaled flag word; | 256: q_bstride == 0, one query
 * set shared by every batch, which the plan cannot see otherwise). */
int m324_attention_plan(int B, int H, int Lq, int Lk, int flags, int dtype, char* buf, int n);
/*   lse (optional, [B,H,Lq] fp32): log2-domain log-sum-exp of every score row, saved for the backward pass. */

/* ------------------------------------------------------------------------------------------
 * m324_attention_merge: one softmax attended in two or three disjoint key parts -> the attention over all keys.
 *   O_i [B*Lq, ldp] (token-major, H*64 columns) and lse_i [B,H,Lq] are what m324_attention left for part i (O2 / lse2 may both
 *   be NULL); O = sum_i 2^(lse_i - m) O_i / sum_i 2^(lse_i - m).  The frame-parallel global blocks (reference
 *   model/Pcd_motion.py:401-405 on a clip sharded over ranks, scripts/4D_from_existing.sh:54-64) attend to the rank's own keys
 *   while the other ranks' k|v rows are still in flight and merge afterwards.
 * ------------------------------------------------------------------------------------------ */
int m324_attention_merge(const void* O0, const float* lse0, const void* O1, const float* lse1, const void* O2, const float* lse2,
                         long ldp, void* O, long ldo, int B, int H, int Lq, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_patchify: video frames -> DINOv2 patch rows.
 *   replaces: permute + F.interpolate(bilinear, align_corners=False) (model/Pcd_motion.py:470-472),
 *             ImageNet normalisation (model/image_encoder/dinov2.py:78-80) and the im2col of the
 *             k=s=14 patch convolution (model_dino.py:160-170).
 *   video [F, Hin, Win, 3] fp32 in [0,1]; out [F * g*g, Kp] (dtype), g = size / patch, column
 *   c*patch*patch + ky*patch + kx (the Conv2d weight's flattening), columns 3*patch*patch..Kp-1 zero.
 * ------------------------------------------------------------------------------------------ */
int m324_patchify(const float* video, int F, int Hin, int Win, int size, int patch,
                  void* out, int Kp, int dtype, void* stream);
/* The same on BYTE frames (ABI 19): video [F, Hin, Win, 3] uint8 in 0..255 -- what a video decoder delivers; every tap is
 * converted as (float)v / 255.0f (IEEE division), i.e. exactly the caller's `frames.float() / 255.0`
 * (scripts/inference_with_video_mesh.py:364), so the rows equal m324_patchify's on the converted frames bit for bit.
 * The long-video driver uploads a quarter of the bytes (motion324_amd/inference.py). */
int m324_patchify_u8(const unsigned char* video, int F, int Hin, int Win, int size, int patch,
                     void* out, int Kp, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_point_encode: Fourier features of points, PointEmbed.embed (model/Pcd_motion.py:178-182).
 *   xyz [P,3] fp32 -> out [P, ld] (dtype): [sin(xyz . basis) (24) | cos (24) | xyz (3) | zeros to 64];
 *   basis = 2^j * pi, j = 0..7 per axis (:164-173).  Always evaluated in fp32 (the reference's bf16
 *   einsum loses the phase).
 * m324_point_concat: writes [normal | rgb | zeros] into columns C..Kp-1 of feat [P, Kp] whose first C
 *   columns already hold point_embed.mlp's output -- the torch.cat at Pcd_motion.py:459,551-553.
 * ------------------------------------------------------------------------------------------ */
int m324_point_encode(const float* xyz, int P, void* out, long ld, int dtype, void* stream);
int m324_point_concat(const float* normal, const float* rgb, int P, void* feat, int C, int Kp,
                      int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_dino_cls_rows: x[f*rows_per_frame + 0, :] = cls + pos[0]  (model_dino.py:127-131).
 * ------------------------------------------------------------------------------------------ */
int m324_dino_cls_rows(const float* cls, const float* pos0, float* x, int F, int rows_per_frame, int C,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_assemble_tokens: trunk input [B,T,4+K+P,C] fp32 =
 *      LN_in( [special(t) (4) | mesh_feat[b] (K) | LN_dino(dino_x[b,t,1+p]) + pos_embed[t*P+p] (P)] )
 *   replaces: DINOv2 final norm + CLS drop (model_dino.py:645, dinov2.py:99-103), pos-embed add
 *             (model/Pcd_motion.py:477-493), token stacking (:495-507) and transformer_input_layernorm (:509).
 *   dino_x [B*T, 1+P, C] fp32 (pre final-norm); dino_w/dino_b final-norm affine, eps_dino (1e-6);
 *   pos [T*P, C]; sp0/spr [4,C]; mesh [B,K,C]; ln_w [C], eps_in (1e-5).  ln_w == NULL: the un-normalised
 *   concatenation is written instead (needed by the backward of transformer_input_layernorm).
 *   drop_p > 0 (training): pos_drop (model/Pcd_motion.py:369-370,490) on the video rows before stacking --
 *   element i = ((b*T+t)*P+p)*C+c of the reference's x is kept iff
 *   (splitmix64(i * 0xD1342543DE82EF95 + drop_seed) >> 40) >= (uint32)(drop_p * 2^24) and scaled by 1/(1-drop_p);
 *   the same (drop_p, drop_seed) must be passed when the rows are recomputed for the backward.
 * ------------------------------------------------------------------------------------------ */
int m324_assemble_tokens(const float* dino_x, const float* dino_w, const float* dino_b, float eps_dino,
                         const float* pos, const float* sp0, const float* spr, const float* mesh,
                         const float* ln_w, float eps_in, float* out,
                         int B, int T, int K, int P, int C, float drop_p, unsigned long long drop_seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_linear_n3: out[M,3] fp32 = A[M,K] . W[3,K]^T + bias -- the xyz regression head's last layer
 *   (shared_mlp_output.3, model/Pcd_motion.py:340,561).  W, bias fp32; A in `dtype`.
 * ------------------------------------------------------------------------------------------ */
int m324_linear_n3(const void* A, long lda, const float* W, const float* bias, float* out,
                   int M, int K, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_mse: *out = weight * mean((pred - target)^2) over n fp32 elements -- model/loss.py:59-61.
 *   `partial` is a caller-provided fp32 scratch of >= 1024 elements.
 * ------------------------------------------------------------------------------------------ */
int m324_mse(const float* pred, const float* target, long n, float weight, float* partial, float* out,
             void* stream);

/* ------------------------------------------------------------------------------------------
 * m324_smooth_trajectories: the caller-side jitter filter applied to pcd_moved, on the GPU.
 *   replaces: smooth_trajectories(method='combined'|'threshold'|'gaussian') (utils/inference_utils.py:99-148),
 *   a Python triple loop over B*N*3 on the CPU in the reference.
 *   trajs/out [B,T,N,3] fp32, tmp same size (scratch).  threshold < 0 skips the threshold pass, sigma <= 0 skips
 *   the gaussian pass (scipy.ndimage.gaussian_filter1d semantics: truncate 4, mode 'nearest').
 * ------------------------------------------------------------------------------------------ */
int m324_smooth_trajectories(const float* trajs, float* tmp, float* out, int B, int T, int N,
                             float threshold, float sigma, void* stream);
/* The two remaining methods of the same caller-side function (utils/inference_utils.py:148-195):
 *   m324_smooth_savgol : method='savgol' -- scipy.signal.savgol_filter(window, polyorder, mode='nearest') = a symmetric
 *     FIR along t with clamped borders; `coef` is a DEVICE array of `window` (odd) fp64 Savitzky-Golay coefficients the host
 *     computes (motion324_amd/postprocess.py); accumulation in fp64, like scipy's convolve1d.  T >= window.
 *   m324_smooth_oneeuro: method='oneeuro' -- OneEuroFilter (:58-97) per scalar coordinate, sequential in t, fp64 state.
 * trajs/out [B,T,N,3] fp32, distinct buffers. */
int m324_smooth_savgol(const float* trajs, float* out, int B, int T, int N, const double* coef, int window, void* stream);
int m324_smooth_oneeuro(const float* trajs, float* out, int B, int T, int N, float mincutoff, float beta, float dcutoff,
                        void* stream);
/* index[i] = argmin_j |query[i] - ref[j]|^2 (lowest j on ties): the vertex-colour assignment of the caller's pre-step
 * (scripts/inference_with_video_mesh.py:112-115, a scipy cKDTree query in the reference).  query [n_query,3], ref [n_ref,3]. */
int m324_nearest_point(const float* query, int n_query, const float* ref, int n_ref, int* index, void* stream);

/* ==========================================================================================
 * Training-side entry points (backward of the path; reference: torch autograd over the same modules,
 * train.py:150-166).  The backward GEMMs reuse m324_gemm on transposed operands:
 *   dA[M,K] = dC[M,N] . Wt[K,N]^T          (Wt = m324_transpose(W))
 *   dW[N,K] (+)= dCt[N,Mp] . At[K,Mp]^T    (dCt, At = m324_transpose of the activations, Mp = round_up(M,64))
 * ========================================================================================== */
/* out[c][r] = in[r][c] for r < rows, zeros for rows <= r < rows_pad.  out is [cols, ld_out >= rows_pad]. */
int m324_transpose(const void* in, long ld_in, void* out, long ld_out, int rows, int cols, int rows_pad,
                   int dtype, void* stream);
/* out[c] (+)= sum over rows of x[r][c] (fp32 accumulation): bias gradients. */
int m324_colsum(const void* x, long ld, float* out, int rows, int cols, int dtype, int accumulate,
                float* scratch, int scratch_rows, void* stream);
/*   scratch (optional, scratch_rows x cols floats): lets tall inputs be reduced by row chunks in parallel. */
/* Many fp32 column sums in ONE launch (ABI 20): the split-K partials of the weight gradients, the per-workgroup partials of
 * the LayerNorm / RMSNorm weight gradients -- ~350 m324_colsum launches of a training step (train.py:166) become ~40.
 *   items[i]: dst[c] (+)= sum over rows of src[r * ld + c], c < cols.  chain = 1: the item CONTINUES item i - 1 (same dst, same
 *   cols): its sum is added to the running value by the same thread right behind it -- exactly what consecutive m324_colsum
 *   calls with accumulate = 1 compute, in the same order (deterministic; bit-identical to them for rows <= 64).
 *   Up to 64 items per launch (the table travels in the kernel arguments; longer lists are cut into several launches). */
typedef struct {
    float* dst;
    const float* src;
    long ld;
    int rows, cols;
    int accumulate;
    int chain;
} m324_colsum_item;
int m324_colsum_multi(const m324_colsum_item* items, int n, void* stream);
/* h = gelu_erf(z);  dz = dh * gelu'(z)  (nn.GELU, transformer.py:58), elementwise over n values. */
int m324_gelu(const void* z, void* h, long n, int dtype, void* stream);
int m324_gelu_bwd(const void* z, const void* dh, void* dz, long n, int dtype, void* stream);
/* LayerNorm backward (weight-only or with bias; the statistics are recomputed from x):
 *   dx[in_row(r)] (+)= d/dx ; the launch has n_partial workgroups of 8 waves (one row per wave at a time) and
 *   partial[n_partial][2C] receives every workgroup's sums of dy*xhat | dy -- reduce with
 *   m324_colsum(partial, 2C, ..., rows = n_partial) to get dw | db (fixed order = deterministic). */
int m324_layernorm_bwd(const float* x, long ldx, const float* w, float eps, const void* dy, long ldy, int dy_dtype,
                       float* dx, long lddx, int accumulate, float* partial, int n_partial, int rows, int C,
                       int gin, int gout, int off, void* stream);
/* The same, and in the same pass: dx_bf16[in_row(r)] = the resulting dx row rounded to bf16 (what the backward GEMMs behind it
 * read -- the reference's autograd hands the fp32 gradient to a bf16 autocast Linear the same way), and partial gets a THIRD
 * block: partial[n_partial][3C] = dy*xhat | dy | column sums of the rounded dx (the bias gradient of the Linear whose output
 * gradient this dx is).  Replaces m324_cast + m324_colsum over the tensor after every LayerNorm backward of the training step. */
int m324_layernorm_bwd_cast(const float* x, long ldx, const float* w, float eps, const void* dy, long ldy, int dy_dtype,
                            float* dx, long lddx, int accumulate, float* partial, int n_partial, int rows, int C,
                            int gin, int gout, int off, void* dx_bf16, long ldc, void* stream);

/* out = in converted between fp32 / bf16 (rows x cols, leading dims in elements). */
int m324_cast(const void* in, long ld_in, int in_dtype, void* out, long ld_out, int out_dtype, int rows, int cols,
              void* stream);
/* Attention backward (flash-attention backward of transformer.py:134-139,209-214 under autograd).
 *   m324_attention_delta: D[B,H,L] = rowsum(dO * O) over each head's 64 columns (token-major O, dO, ld >= H*64).
 *   m324_attention_bwd:   Qs[Bq,H,Lq,64] (pre-scaled q, q_bstride 0 = shared), K, V [B,H,Lk,64] row-major, dO [B,H,Lq,64]
 *                         head-major, lse / D [B,H,Lq] -> dQ [B,H,Lq,64] (w.r.t. the UNSCALED normalised q; per batch even when
 *                         q is shared: sum over batches afterwards), dK, dV [B,H,Lk,64].  fp32 arithmetic. */
int m324_attention_delta(const void* O, const void* dO, long ld, float* D, int B, int H, int L, int dtype, void* stream);
int m324_attention_bwd(const void* Qs, long q_bstride, const void* K, const void* V, const void* dO, const float* lse,
                       const float* D, void* dQ, void* dK, void* dV, int B, int H, int Lq, int Lk, float scale,
                       int dtype, void* stream);
/* bf16 MFMA implementation of the same backward (two kernels: dQ per query block, dK/dV per key block, both
 * recomputing the scores).  Besides the row-major operands it takes the transposed, permuted copies m324_qkv_split
 * emits: Qst [Bq,H,64,Lqp], Kt [B,H,64,Lkp], dOt [B,H,64,Lqp] (qt_bstride / q_bstride = 0 for a shared query set). */
int m324_attention_bwd_mfma(const void* Qs, const void* Qst, long q_bstride, long qt_bstride, const void* K, const void* Kt,
                            const void* V, const void* dO, const void* dOt, const float* lse, const float* D,
                            void* dQ, void* dK, void* dV, int B, int H, int Lq, int Lk, float scale, void* stream);
/* Backward of m324_qkv_split: head-major dQ/dK/dV -> token-major gradients of the projections (RMSNorm backward for
 * q, k when q_w / k_w are given; raw projections needed).  partial [n_partial][128]: per-block sums of the q_norm | k_norm
 * weight gradients, reduce with m324_colsum. */
int m324_qkv_split_bwd(const void* dQ, const void* dK, const void* dV, const void* q_raw, long ldq, const void* k_raw,
                       long ldk, const float* q_w, const float* k_w, float eps, void* dq_out, long ldoq, void* dk_out,
                       long ldok, void* dv_out, long ldov, float* partial, int n_partial, int B, int L, int H, int dtype,
                       void* stream);
/* Backward of m324_linear_n3: dA[M,K] = dout[M,3] . W[3,K]; partial [n_partial][3*K] = per-block sums of dout^T A
 * (reduce with m324_colsum to get dW).  mul (optional, ABI 22; A's dtype, [M, ldmul]): dA[m,k] is multiplied by mul[m,k] --
 * with the gelu'(z) the head's first Linear left (M324_AUX_STORE_GELU_GRAD) the kernel delivers d(pre-activation) directly and
 * the m324_gelu_bwd pass over the tensor is gone. */
int m324_linear_n3_bwd(const void* A, long lda, const float* W, const float* dout, void* dA, long ldda, float* partial,
                       int n_partial, int M, int K, int dtype, const void* mul, long ldmul, void* stream);
/* d[i] = coef * (*grad_scale) * (pred[i] - target[i])  -- backward of m324_mse (coef = 2 * weight / n). */
int m324_mse_bwd(const float* pred, const float* target, const float* grad_scale, float coef, float* d, long n,
                 void* stream);
/* torch.optim.AdamW step on one tensor (utils/training_utils.py:38-52: betas (0.9, 0.95), eps 1e-8, decoupled decay);
 * grad_scale: optional device scalar multiplying the gradient (clipping coefficient, train.py:196). */
int m324_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, const float* grad_scale, void* stream);
/* The same update over the optimizer's WHOLE flat parameter buffer in one launch (multi-tensor semantics of
 * torch.optim.AdamW(fused=True), utils/training_utils.py:52): elements [0, n_decay) take `weight_decay` (parameters with
 * dim() > 1, training_utils.py:39-47), elements [n_decay, n) take none.  n, n_decay multiples of 4; 16-byte aligned. */
int m324_adamw_flat(float* p, const float* g, float* m, float* v, long n, long n_decay, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, const float* grad_scale, void* stream);
/* bf16 copies of the optimizer's flat fp32 parameter buffer in ONE launch (ABI 21): for every item (a Linear weight [rows, cols], cols a
 * multiple of 64, src_off -- a multiple of 4 -- elements into `src`) the row-major copy at dst + dst_off (the forward GEMM's operand) and, when dstT_off >= 0, the
 * transposed copy [cols, ldT] at dstT + dstT_off (the dgrad GEMM's operand; columns [rows, ldT) receive nothing but zeros: the caller
 * zeroes the buffer once).  Round to nearest even, as torch's .to(bfloat16).  items_dev: DEVICE memory (the table is constant
 * between optimizer steps), ordered by first_tile = the number of 64 x 64 tiles in front of the item; n_tiles = their total.
 *   replaces: the per-step torch casts + m324_transpose launches of every weight after optimizer.step()
 *             (train.py:203-213: the reference's autocast re-casts each weight inside every Linear call instead). */
typedef struct m324_mirror_item {
    long src_off, dst_off, dstT_off;
    long first_tile;
    int rows, cols, ldT, pad_;
} m324_mirror_item;
int m324_weight_mirror(const float* src, void* dst, void* dstT, const m324_mirror_item* items_dev, int n_items, long n_tiles,
                       void* stream);
/* *out (+)= sum g^2 (gradient-norm pieces); sanitize != 0 first applies nan_to_num(0, 1e-6, -1e-6) in place
 * (train.py:181-183).  partial: >= 1024 floats of scratch. */
int m324_grad_sumsq(float* g, long n, int sanitize, float* partial, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Collectives of the path over RCCL / xGMI (one process per GPU).
 *   replaces: the NCCL collectives torch.distributed issues for the reference -- DDP's gradient all-reduce
 *             (train.py:88-89,159-166; setup.py:134-140 creates the process group) -- and carries the two exchanges of the
 *             multi-GPU inference modes (k|v all-gather per global block, final [T, N, 3] all-gather).
 *   The Python host of this repo keeps using torch.distributed ("nccl" IS RCCL on ROCm) because the reference's callers
 *   own that process group; these entry points give a torch-free host the same collectives.  RCCL is resolved when
 *   m324_comm_init is first called (symbols already in the process, else librccl.so): libm324.so itself does not link it.
 *   Bootstrap: rank 0 calls m324_comm_unique_id and hands the 256-character hex string to the other ranks (file, env,
 *   socket -- the caller's business); every rank then calls m324_comm_init on ITS device (hipSetDevice first).
 *   All calls only enqueue on `stream`; buffers are device pointers; dtype M324_F32 / M324_BF16.
 * ------------------------------------------------------------------------------------------ */
typedef struct m324_comm m324_comm;
int m324_comm_unique_id(char* hex, int n);                         /* n >= 257 */
int m324_comm_init(m324_comm** comm, const char* unique_id_hex, int rank, int world);
/* in place; average != 0 divides by the number of ranks (DDP's gradient mean) */
int m324_comm_allreduce(m324_comm* comm, void* buf, long count, int dtype, int average, void* stream);
/* recv[r * count_per_rank ...] = rank r's send (rank order = frame order in the frame-parallel forward) */
int m324_comm_allgather(m324_comm* comm, const void* send, void* recv, long count_per_rank, int dtype, void* stream);
int m324_comm_destroy(m324_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* M324_H */

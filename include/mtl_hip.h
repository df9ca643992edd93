/* mtl_hip.h -- C ABI of libmtl_hip.so: the MI355X (gfx950) kernels of the meta-transfer-learning hot path.
 *
 * Drop-in boundary.  The reference (audioku/meta-transfer-learning) has no FFI layer: its
 * hot path is Python calling stock PyTorch ops.  This library supplies, as hand-written HIP, every op that
 * path issues; each entry point cites the reference call site(s) it replaces (paths relative to the reference
 * repository root).  The Python host layer (meta-transfer-learning_amd/) binds these with ctypes and keeps the
 * reference's own Python signatures (Transformer.forward, TransientTrainer.train, ...).  INTEGRATION.md shows
 * the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless named `*_host`; the library never
 *     allocates, frees or retains memory.  Scratch space is passed in (`workspace`, size from *_workspace()).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No call synchronises the device.
 *   - return 0 on success, a negative errno-style code otherwise (-22 bad argument, -5 launch failure);
 *     nothing throws or aborts.  Re-entrant; one host thread per device.
 *   - all tensors fp32 row-major unless stated; token ids int64 (`long`), lengths int32.
 *   - activations inside the conv stack are channels-last with TIME outermost: (B, T, F, C).
 *   - "accum" outputs are ADDED to (gradient accumulation semantics of .backward()).
 *   - all reductions are fixed-order (no floating-point atomics): results are run-to-run deterministic.
 */
#ifndef MTL_HIP_H
#define MTL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int mtl_abi_version(void);

/* ---- GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) -------------------------------------------
 * C[z] = epilogue( alpha * opA(A[z]) . opB(B[z]) ),  z = 0..batch-1,  operand z at  base + (z/H)*s?b + (z%H)*s?h
 *   transA = 0: A is M x K (lda)   | 1: A is stored K x M (lda)
 *   transB = 0: B is K x N (ldb)   | 1: B is stored N x K (ldb)
 *   epilogue: + bias[n] (nullable) ; ReLU if flags&MTL_GEMM_RELU ; zero where gate[m*ldg+n] <= 0 (nullable,
 *             same batch offsets as C) ; += C if flags&MTL_GEMM_ACCUM.
 *   sBias: bias stride (floats) of the outer batch index b (0 = one bias vector for every item) -- lets the three Q/K/V
 *             projections of an attention block, whose parameters sit at a constant stride in the flat buffer, run as ONE call.
 *   workspace (nullable): when the output has too few tiles (x batch items) to fill the 256 CUs, K is split over the grid
 *             into workspace slabs that a second kernel sums in fixed order (deterministic split-K).
 * Replaces nn.Linear forward/backward (modules/encoder.py:72; modules/common_layers.py:130,287-289,303;
 * modules/decoder.py:108-110) and torch.bmm (modules/common_layers.py:321,329) incl. the permute/contiguous
 * copies at common_layers.py:291-293,301 (heads are addressed by stride instead). */
#define MTL_GEMM_RELU 1
#define MTL_GEMM_ACCUM 2
int mtl_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                 const float* B, int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags,
                 int batch, int H, long sAb, long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, float* workspace,
                 long workspace_bytes);

/* mtl_gemm_f32 with two extensions, and a small-tile engine (32 x 32 tiles on v_mfma_f32_16x16x4_f32, no workspace, no second
 * launch) for products whose 64 x 64 tiling would leave most of the 256 CUs idle (< 192 tiles):
 *   kbatch > 1: C = epilogue( alpha * sum_{z<kbatch} opA(A + z*sAk) . opB(B + z*sBk) ) -- the K loop runs over the z items inside
 *               ONE launch (dx of the Q/K/V low-rank a-stages: three serialised accumulate launches before);
 *   rowsum (nullable, transA only): rowsum[(z/H)*sRowsum + m] += sum_k opA(A)[m][k] -- the bias gradient colsum(dy) of
 *               dW = dy^T . x as a by-product of the weight-gradient product (fixed-order reduction).
 * Products that fill the chip (and the few-tile very-long-K ones) are forwarded to mtl_gemm_f32 unchanged. */
int mtl_gemm_f32_ex(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags,
                    int batch, int H, long sAb, long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk,
                    long sBk, float* rowsum, long sRowsum, float* workspace, long workspace_bytes,
                    long sBiasH, long sRowsumH /* bias / rowsum strides of the INNER batch index (z % H); sBias / sRowsum belong to z / H:
                                                  e.g. the K and V projections (inner) of all decoder layers (outer) in one launch */);
/* mtl_gemm_f32_ex with a THIRD, outermost batch level -- the tasks of a meta-step (trainer/asr/transient_trainer.py:178-237: every
 * task's passes are independent given theta0) batched into one launch: batch = tasks * (items per task); item z belongs to task
 * zt = z / (batch / tasks) and its operands sit at base + zt * s?t + (the two-level offsets of z % (batch / tasks)).  The weights
 * of the validation passes differ per task (theta'_t = theta0 - alpha g_t, a stack with stride sBt; 0 for the shared theta0 of
 * the training passes), per-task gradients accumulate into a stack (sCt / sRowsumT). */
int mtl_gemm_f32_tb(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, const float* gate, int ldg, int flags,
                    int batch, int H, long sAb, long sAh, long sBb, long sBh, long sCb, long sCh, long sBias, int kbatch, long sAk,
                    long sBk, float* rowsum, long sRowsum, float* workspace, long workspace_bytes, long sBiasH, long sRowsumH,
                    int tasks, long sAt, long sBt, long sCt, long sBiasT, long sRowsumT);
/* which engine mtl_gemm_f32_ex / _tb picks (16-byte aligned operands, not transA && transB): 2 = bf16-split engine
 * (gemm_x3_kernel<...>, below -- also its split-K form: a product with fewer tiles than the engine's threshold and K >= 4096 that is
 * called with a workspace runs as equal K slices (batched launch into the workspace) + x3_splitk_sum_kernel, fixed order),
 * 1 = small-tile (gemm16_kernel<...>), 0 = forwarded to mtl_gemm_f32 (gemm_kernel<...>); used by bench.py to attribute launch
 * timings to the rocprofv3 kernel classes */
int mtl_gemm_f32_ex_route(int M, int N, int K, int batch, int kbatch, int has_rowsum);
/* Large products of mtl_gemm_f32_ex / _tb run on the bf16 matrix pipe: every fp32 operand element is split EXACTLY into three bf16
 * pieces on its way to LDS and a block product is six v_mfma_f32_32x32x16_bf16 accumulated in fp32 (a0 b0 + a0 b1 + a1 b0 + a1 b1 +
 * a0 b2 + a2 b0; dropped terms < 2^-23 |a||b|): fp32-class results (tests/test_ops_gpu.py compares both engines with fp64), no
 * scaling and no range caveat, 2.67 x the rate of the fp32 MFMA.  A product is routed there when its grid of 256 x 128 output tiles
 * (x batch items) has at least `min_tiles` workgroups and its operands are 16-byte aligned (csrc/mtl_gemm_x3.hip; nn.Linear forward
 * / backward of modules/common_layers.py:130,287-289,303 and modules/decoder.py:109 in a task-batched pass).
 * mtl_gemm_x3_min_tiles(set): set >= 0 replaces the threshold (0 = engine off; default 128 or the MTL_GEMM_X3 environment
 * variable), set < 0 only queries; returns the previous value. */
int mtl_gemm_x3_min_tiles(int set);

/* Task-batched two-piece fp16 product on the tile engine of csrc/mtl_gemm_x3.hip (256 x 128 x 32 tiles, 8 waves, split interleaved
 * with the MFMAs; 3 v_mfma_f32_32x32x16_f16 per step): for task t < tasks
 *   C_t[M,N] = A_t[M,K] . op(B_t) (+ bias_t[N]) (gate: C = gate_t[m][n] > 0 ? C : 0),   op(B) = B[N,K]^T (transB = 1) or B[K,N] (transB = 0),
 * operands of task t at base + t * s?t (sBt = 0: shared weights), its bounds max|A_t| / max|B_t| at amax_? + t * sAmax? floats
 * (MTL_AMAX_SLOTS slot heads each, as for the *_h2 convolutions; gate shares C's offsets).  The encoder's input Linear (5120 -> 512,
 * models/asr/transformer.py:136-140, modules/encoder.py:72) of all tasks of a pass in one launch, and its data gradient straight from the
 * un-transposed weight (transB = 0).  Any M, N, K; 16-byte aligned operands, lda / ldb / strides multiples of 4.
 * workspace (nullable): with ONE task, few output tiles and K >= 2048 the reduction is split over the grid into it (fixed-order sum). */
int mtl_gemm_h2_tb(void* stream, int transB, int M, int N, int K, const float* A, int lda, const float* amax_a, long sAmaxA,
                   const float* B, int ldb, const float* amax_b, long sAmaxB, float* C, int ldc, const float* bias, const float* gate,
                   int ldg, int tasks, long sAt, long sBt, long sCt, long sBiasT, float* workspace, long workspace_bytes);
/* the transposed-A form of the same engine: C_t[M,N] = A_t[K,M]^T . B_t[K,N] on two fp16 pieces (bounds as above) -- the weight gradient
 * of the encoder's input Linear, dW = de0^T . p2 (K = rows of the pass, M = 512, N = 5120), whose operands' bounds the pass has anyway */
int mtl_gemm_h2_tn_tb(void* stream, int M, int N, int K, const float* A, int lda, const float* amax_a, long sAmaxA, const float* B, int ldb,
                      const float* amax_b, long sAmaxB, float* C, int ldc, int tasks, long sAt, long sBt, long sCt);

/* ---- VGG front-end: models/asr/transformer.py:48-59 (Conv2d 3x3 s1 p1 + ReLU [+ MaxPool2d(2,2)]) ------
 * x_ref is the reference's (B,1,F,T) input; everything downstream is (B,T,F,C). */
/* amax_y (optional): MTL_AMAX_FLOATS floats, slot heads atomically raised to max(y) -- the `amax_x` of a following *_h2 convolution; zero them first. */
int mtl_conv0_relu_fwd(void* stream, const float* x_ref, const float* w /*(64,1,3,3)*/, const float* bias, float* y,
                       int B, int T, int F, float* amax_y);
long mtl_conv0_wgrad_workspace(void);
int mtl_conv0_wgrad(void* stream, const float* x_ref, const float* dy, float* dw /*accum*/, float* db /*accum*/,
                    float* workspace, int B, int T, int F);
/* conv0 and the conv bias column sums for the batches of `tasks` meta-tasks in one launch (pair) each -- the task-batched passes issue
 * them once per phase instead of once per task (8 launches of 8 samples: 593 / 527 / 132 us, one launch: 523 / 424 / 55 us).  Task k reads
 * its input batch at x + k sX floats (0: all tasks see the same batch, the shared validation batch), its weights / bias / output bound at
 * + k sW / sBias / sAmax floats; y and dy hold tasks * B samples; gradients accumulate onto dw + k sDw, db + k sDb.  Per task bitwise the
 * single-task calls for the forward; the weight gradient shares the 1024 partial rows of its workspace among the tasks. */
int mtl_conv0_relu_fwd_tb(void* stream, const float* x, const float* w, const float* bias, float* y, int B, int T, int F, float* amax_y,
                          int tasks, long sX, long sW, long sBias, long sAmax);
int mtl_conv0_wgrad_tb(void* stream, const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int T, int F, int tasks,
                       long sX, long sDw, long sDb);
/* column sums of `tasks` contiguous row blocks (rows x cols each) in one launch pair; workspace: tasks x mtl_colsum_workspace(rows, cols)
 * bytes (+ 16 per task); cols = 4 x 2^j <= 256, 16-byte aligned X */
int mtl_colsum_accum_tb(void* stream, const float* X, long rows, int cols, float* out, float* workspace, float* amax, int tasks, long sOut,
                        long sAmax);
/* (Cout,Cin,3,3) -> w_fwd [9][Cin][Cout] and w_dgrad [9][Cout][Cin] (taps rotated 180 degrees) */
int mtl_conv3x3_wprep(void* stream, const float* w_ref, float* w_fwd, float* w_dgrad, int Cout, int Cin);
int mtl_conv3x3_relu_fwd(void* stream, const float* x, const float* w_fwd, const float* bias, float* y, int B, int T,
                         int F, int Cin, int Cout);
/* fused conv + bias + ReLU + 2x2 max-pool (floor mode; first maximum in torch's window order wins):
 * p_out (B,T/2,F/2,Cout), argmax u8 in {0..3} = 2*(f&1) + (t&1).  The un-pooled activation is never written. */
int mtl_conv3x3_relu_pool_fwd(void* stream, const float* x, const float* w_fwd, const float* bias, float* p_out,
                              unsigned char* argmax, int B, int T, int F, int Cin, int Cout);
/* dx = conv_transpose(dy) gated by ReLU of the forward input activation `act` (same shape as dx).
 * argmax != NULL: dy is the POOLED gradient (B,T/2,F/2,Cout); un-pooling is fused into the operand load.
 * T,F,Cin,Cout describe the FORWARD convolution. */
int mtl_conv3x3_dgrad(void* stream, const float* dy, const unsigned char* argmax, const float* w_dgrad,
                      const float* act, float* dx, int B, int T, int F, int Cin, int Cout);
/* Split-bf16 ("x3") variants of the three calls above: identical semantics and fp32-class error, computed with six
 * v_mfma_f32_32x32x16_bf16 per 16-deep step on exact 3-way bf16 splits of both operands (2.67x the fp32 MFMA roof).
 * w3_fwd / w3_dgrad: bf16 [3][K-tile][rows][32] buffers (3 * 9*Cin*Cout * 2 bytes each) from mtl_conv3x3_wprep_x3; the four
 * 16-byte chunks of a row are stored at chunk ^ ((row >> 2) & 3) (the LDS image the kernels DMA with global_load_lds). */
int mtl_conv3x3_wprep_x3(void* stream, const float* w_ref, void* w3_fwd, void* w3_dgrad, int Cout, int Cin);
int mtl_conv3x3_relu_fwd_x3(void* stream, const float* x, const void* w3_fwd, const float* bias, float* y, int B, int T, int F,
                            int Cin, int Cout);
int mtl_conv3x3_relu_pool_fwd_x3(void* stream, const float* x, const void* w3_fwd, const float* bias, float* p_out,
                                 unsigned char* argmax, int B, int T, int F, int Cin, int Cout);
int mtl_conv3x3_dgrad_x3(void* stream, const float* dy, const unsigned char* argmax, const void* w3_dgrad, const float* act,
                         float* dx, int B, int T, int F, int Cin, int Cout);
/* Two-piece fp16 ("h2") variants: the same calls on 2-way fp16 splits of both operands -- three v_mfma_f32_32x32x16_f16 per
 * 16-deep step (half the matrix work of x3, twice its roof), fp32 accumulation, error ~1.5x that of an fp32 convolution
 * (h l' + l h' + h h' keeps 22 significand bits).  fp16 has 5 exponent bits, so every operand tensor comes with a DEVICE
 * BOUND `amax_*` >= max|tensor| (an upper bound within a few powers of two is as good): the kernels scale by the power of two
 * that puts the bound into [2^14, 2^15) before splitting and un-scale the accumulators exactly.  A bound is MTL_AMAX_FLOATS
 * floats: the maximum over its 64 slot heads counts (producers raise slot (workgroup % 64) atomically -- L2 serialises atomics
 * per cache line, one hot line costs thousands of them 40-300 us).  Producers deliver the bounds for free: `amax_y` / `amax_p` / `amax_dx` (optional outputs; zero all slots first),
 * mtl_conv0_relu_fwd's amax_y, mtl_colsum_accum's amax (the bias-gradient pass reads the whole gradient anyway; it WRITES all
 * slots), or mtl_absmax_f32.
 * w2_fwd / w2_dgrad: mtl_conv3x3_wprep_h2_bytes() each: [2][K-tile][rows][32] fp16 in the x3 layout + the fp32 weight scale. */
#ifndef MTL_AMAX_SLOTS
#define MTL_AMAX_SLOTS 64   /* every amax_* argument is MTL_AMAX_FLOATS floats: 64 slots, one 128-byte line apart (slot i at */
#define MTL_AMAX_STRIDE 32  /* [i * 32]); the bound is the maximum over the slot heads, the other floats are padding          */
#define MTL_AMAX_FLOATS (MTL_AMAX_SLOTS * MTL_AMAX_STRIDE)
#endif
long mtl_conv3x3_wprep_h2_bytes(int Cout, int Cin);
int mtl_conv3x3_wprep_h2(void* stream, const float* w_ref, void* w2_fwd, void* w2_dgrad, int Cout, int Cin);
/* the same for n <= 3 layers with one call (the pass prepares conv2 / conv5 / conv7 together); unused triples are ignored */
int mtl_conv3x3_wprep_h2_batch(void* stream, int n, const float* w0, void* f0, void* d0, int Cout0, int Cin0, const float* w1, void* f1,
                               void* d1, int Cout1, int Cin1, const float* w2, void* f2, void* d2, int Cout2, int Cin2);
/* ... and for `sets` parameter sets (the theta' stack of a task-batched validation pass) with the same two launches: set z reads its
 * weights at w_i + z sSrc floats and writes its prepared blocks at f_i / d_i + z sDst_i BYTES (multiples of 4) */
int mtl_conv3x3_wprep_h2_batch_tb(void* stream, int n, const float* w0, void* f0, void* d0, int Cout0, int Cin0, const float* w1, void* f1,
                                  void* d1, int Cout1, int Cin1, const float* w2, void* f2, void* d2, int Cout2, int Cin2, int sets,
                                  long sSrc, long sDst0, long sDst1, long sDst2);
int mtl_conv3x3_relu_fwd_h2(void* stream, const float* x, const float* amax_x, const void* w2_fwd, const float* bias, float* y,
                            float* amax_y, int B, int T, int F, int Cin, int Cout);
int mtl_conv3x3_relu_pool_fwd_h2(void* stream, const float* x, const float* amax_x, const void* w2_fwd, const float* bias,
                                 float* p_out, unsigned char* argmax, float* amax_p, int B, int T, int F, int Cin, int Cout);
int mtl_conv3x3_dgrad_h2(void* stream, const float* dy, const float* amax_dy, const unsigned char* argmax, const void* w2_dgrad,
                         const float* act, float* dx, float* amax_dx, int B, int T, int F, int Cin, int Cout);
/* The same three launches for the batches of `tasks` meta-tasks at once (trainer/asr/transient_trainer.py:178-237: the tasks of a
 * meta-step are independent given theta0): x / y / dy / dx / act / argmax hold tasks * B samples, task k = samples [k B, (k + 1) B);
 * task k reads its prepared weights at w2 + k * sW BYTES (0: all tasks share theta0's -- the training passes), its bias at
 * bias + k * sBias floats, its operand bound at amax + k * sAmaxX floats and raises its own output bound at amax_y + k * sAmaxY.
 * Per task bitwise the single-task launch (same tiles, same scales); one launch of 64 samples costs 2-14 % less than eight of 8.
 * widths (optional, `tasks` ints on the device, with mtl_zero_tails): the tasks' own frame counts when their batches were stacked at a
 * common T; rows t >= widths[k] >> wshift of task k's samples (t = the row index of this launch's convolution: before its pooling, the
 * data gradient's output rows) are then NOT computed -- whole pixel-tile rows beyond them are left out of the launch and their outputs
 * stay untouched: the caller clears them (mtl_zero_tails) before anything reads them.  NULL: every row. */
int mtl_conv3x3_relu_fwd_h2_tb(void* stream, const float* x, const float* amax_x, const void* w2_fwd, const float* bias, float* y,
                               float* amax_y, int B, int T, int F, int Cin, int Cout, int tasks, long sW, long sBias, long sAmaxX, long sAmaxY,
                               const int* widths, int wshift);
int mtl_conv3x3_relu_pool_fwd_h2_tb(void* stream, const float* x, const float* amax_x, const void* w2_fwd, const float* bias, float* p_out,
                                    unsigned char* argmax, float* amax_p, int B, int T, int F, int Cin, int Cout, int tasks, long sW,
                                    long sBias, long sAmaxX, long sAmaxP, const int* widths, int wshift);
int mtl_conv3x3_dgrad_h2_tb(void* stream, const float* dy, const float* amax_dy, const unsigned char* argmax, const void* w2_dgrad,
                            const float* act, float* dx, float* amax_dx, int B, int T, int F, int Cin, int Cout, int tasks, long sW,
                            long sAmaxDy, long sAmaxDx, const int* widths, int wshift);
/* weight (+ bias) gradients of `tasks` meta-tasks in one launch: x / dy / argmax hold tasks * B samples; task k reads its bounds at
 * amax_x + k sAmaxX, amax_dy + k sAmaxDy floats and accumulates onto dw_ref + k sDw, db + k sDb floats (the per-task gradient stack).
 * The launch's partial slabs (workspace: mtl_conv3x3_wgrad_x3_workspace, unchanged) are dealt to the tasks in equal contiguous ranges,
 * so one launch writes / reduces as many slabs as ONE single-task launch; tasks <= slabs.  Fixed-order sums: deterministic; the
 * partition of a task's pixels over slabs differs from the single-task launch (same class of rounding, not the same bits). */
int mtl_conv3x3_wgrad_h2_tb(void* stream, const float* x, const float* amax_x, const float* dy, const float* amax_dy,
                            const unsigned char* argmax, float* dw_ref, float* db, float* workspace, long workspace_bytes, int B, int T,
                            int F, int Cin, int Cout, int tasks, long sAmaxX, long sAmaxDy, long sDw, long sDb);
/* The same four several-task launches on the EXACT 3 x bf16 split (MTL_CONV=x3: every fp32 operand bit kept, models/asr/transformer.py:48-59
 * is fp32): no bounds; w3 + k * sW bytes, bias + k * sBias floats, dw_ref + k sDw / db + k sDb floats as above (db nullable: the bias
 * gradient = per-channel sums of dy rides on the weight-gradient launch). */
int mtl_conv3x3_relu_fwd_x3_tb(void* stream, const float* x, const void* w3_fwd, const float* bias, float* y, int B, int T, int F, int Cin,
                               int Cout, int tasks, long sW, long sBias, const int* widths, int wshift);
int mtl_conv3x3_relu_pool_fwd_x3_tb(void* stream, const float* x, const void* w3_fwd, const float* bias, float* p_out, unsigned char* argmax,
                                    int B, int T, int F, int Cin, int Cout, int tasks, long sW, long sBias, const int* widths, int wshift);
int mtl_conv3x3_dgrad_x3_tb(void* stream, const float* dy, const unsigned char* argmax, const void* w3_dgrad, const float* act, float* dx,
                            int B, int T, int F, int Cin, int Cout, int tasks, long sW, const int* widths, int wshift);
int mtl_conv3x3_wgrad_x3_tb(void* stream, const float* x, const float* dy, const unsigned char* argmax, float* dw_ref, float* db,
                            float* workspace, long workspace_bytes, int B, int T, int F, int Cin, int Cout, int tasks, long sDw, long sDb);
/* amax[MTL_AMAX_FLOATS]: slot heads raised so that their maximum is >= max|x[0..n)| (atomic; zero them first) */
int mtl_absmax_f32(void* stream, const float* x, long n, float* amax);
/* the same for `tasks` tensors of n floats at x + k sX, bound k at amax + k sAmax floats, one launch */
int mtl_absmax_f32_tb(void* stream, const float* x, long n, float* amax, int tasks, long sX, long sAmax);
/* Census of an h2 operand against its bound -- the runtime guard of the two-piece fp16 arithmetic (csrc/mtl_h2.h: one power-of-two scale per
 * tensor; an element more than ~2^17.5 below max|tensor| keeps fewer than 22 significand bits).  For `tasks` tensors of n floats at
 * x + k sX with bounds at amax + k sAmax floats, ADDS to counts + k sCounts (three 64-bit counters each; zero them first):
 * [0] non-zero elements, [1] of those, elements that keep fewer than 22 bits, [2] fewer than 16 bits.  Exact (integer atomics).
 * The reference's inputs are normalised per utterance (utils/data_loader.py:84-94), which is why its activations stay in the
 * full-precision regime; TransientTrainer samples this census periodically and leaves h2 when they do not. */
int mtl_h2_census(void* stream, const float* x, long n, const float* amax, unsigned long long* counts, int tasks, long sX, long sAmax,
                  long sCounts);
/* Several tasks' batches of different frame counts in ONE pass, padded to the widest (data.py collate pads every task's batch to its OWN
 * longest utterance, transient_trainer.py:178-237 runs them one by one): y is (n, T, row) floats, sample s belongs to task s / per_task and
 * its frames [widths[task] >> shift, T) are cleared, so that the next convolution meets the zero border of the task's own image and the
 * ReLU gates of the backward hold every gradient out of them.  widths: `n / per_task` ints on the device (full-resolution frame counts;
 * shift = 1 after the first pooling).  row % 4 == 0, y 16-byte aligned. */
int mtl_zero_tails(void* stream, float* y, int n, int T, int row, const int* widths, int shift, int per_task);
long mtl_conv3x3_wgrad_workspace(int B, int T, int F, int Cin, int Cout, int pooled);
/* dw_ref (Cout,Cin,3,3) += sum_pixels x (x) dy ; dy dense (B,T,F,Cout) or pooled + argmax as above. */
int mtl_conv3x3_wgrad(void* stream, const float* x, const float* dy, const unsigned char* argmax, float* dw_ref,
                      float* workspace, long workspace_bytes, int B, int T, int F, int Cin, int Cout);
/* Split-bf16 ("x3") weight gradient: same contract as the two calls above (its own workspace size).  Halo-tiled x staged in
 * LDS, transpose reads (ds_read_b64_tr_b16) for the pixel-major reduction, dy fragments loaded straight into MFMA layout. */
long mtl_conv3x3_wgrad_x3_workspace(int B, int T, int F, int Cin, int Cout, int pooled);
int mtl_conv3x3_wgrad_x3(void* stream, const float* x, const float* dy, const unsigned char* argmax, float* dw_ref,
                         float* workspace, long workspace_bytes, int B, int T, int F, int Cin, int Cout);
/* h2 weight gradient (workspace: mtl_conv3x3_wgrad_x3_workspace); amax_dy bounds the dense OR the pooled gradient it is given.
 * db (nullable, accum): the bias gradient db[c] += sum over pixels of dy -- the kernel's dy loaders see every element anyway, so
 * the separate column-sum pass over dy (164 MB for conv5) is not needed. */
int mtl_conv3x3_wgrad_h2(void* stream, const float* x, const float* amax_x, const float* dy, const float* amax_dy,
                         const unsigned char* argmax, float* dw_ref, float* db, float* workspace, long workspace_bytes, int B, int T,
                         int F, int Cin, int Cout);
/* wp[o][h*C+c] = w[o][c*Hh+h]  (inverse_accum: dst[o][c*Hh+h] += src[o][h*C+c]); the (C*H) flattening of
 * models/asr/transformer.py:136-138 folded into encoder.input_linear's weight instead of an activation copy. */
int mtl_permute_hc(void* stream, const float* src, float* dst, int rows, int C, int Hh, int inverse_accum,
                   float* amax /* nullable, MTL_AMAX_FLOATS: raised to max|src| (zero it first) */);
/* the same for `tasks` weight sets at strides sSrc / sDst (floats) and bounds at sAmax (floats) in ONE launch (the theta' stack) */
int mtl_permute_hc_tb(void* stream, const float* src, float* dst, int rows, int C, int Hh, int inverse_accum, float* amax, int tasks,
                      long sSrc, long sDst, long sAmax);

/* ---- LayerNorm(x + residual) * gamma + beta (+ pe[row % T]) then * keep[row] -----------------------------
 * nn.LayerNorm eps inside the sqrt, biased variance (modules/common_layers.py:131,304; modules/encoder.py:72-73)
 * with the non_pad_mask multiplies of modules/encoder.py:101,104 and modules/decoder.py:314,318,321 fused.
 * d in {64,128,256,512,1024}.  Saves xhat (rows x d) and rstd (rows) for the backward. */
/* xmask (nullable u8 keep-mask of x, from mtl_dropout_mask) / xscale = 1/(1-p): dropout of the sub-layer output BEFORE the
 * residual add (modules/common_layers.py:130,303). */
int mtl_layernorm_fwd(void* stream, const float* x, const float* residual, const float* gamma, const float* beta,
                      const float* pe, const int* keep, const unsigned char* xmask, float xscale, float* y, float* xhat,
                      float* rstd, int rows, int d, int T, float eps);
/* Task-grouped form: the rows are consecutive groups of rows_per_group rows (the batches of the tasks of a meta-step in one
 * launch); group g reads gamma / beta at + g * sParam floats (0: shared parameters). */
int mtl_layernorm_fwd_g(void* stream, const float* x, const float* residual, const float* gamma, const float* beta,
                        const float* pe, const int* keep, const unsigned char* xmask, float xscale, float* y, float* xhat,
                        float* rstd, int rows, int d, int T, float eps, int rows_per_group, long sParam);
long mtl_layernorm_bwd_workspace(int rows, int d);
/* with xmask: dz is the residual-branch gradient and dzm = dz * mask * xscale the sub-layer-branch gradient (dsum sums dzm);
 * dz2 (nullable): a second copy of dz (the residual path accumulates onto it while dz stays intact for the weight gradient) */
int mtl_layernorm_bwd(void* stream, const float* dy, const float* xhat, const float* rstd, const float* gamma,
                      const int* keep, const unsigned char* xmask, float xscale, float* dz, float* dzm /*nullable*/,
                      float* dz2 /*nullable*/, float* dgamma /*accum*/, float* dbeta /*accum*/,
                      float* dsum /*nullable, accum: += column sums of the sub-layer-branch gradient (its linear's bias grad)*/,
                      float* workspace, int rows, int d,
                      int defer_reduce /* 1: leave the per-wave partials in `workspace` (keep it alive, one per instance) and add them
                                          to dgamma / dbeta / dsum later with mtl_ln_param_reduce_batch */);
/* Task-grouped form: a wave's rows belong to one group; group g owns the partial rows [g * W, (g + 1) * W) of `workspace`
 * (W = mtl_layernorm_bwd_g_waves(rows_per_group); 3 * d floats per partial row), reads gamma at + g * sParam and -- unless the
 * reduction is deferred -- adds to dgamma / dbeta / dsum at + g * sGrad. */
int mtl_layernorm_bwd_g_waves(int rows_per_group);
long mtl_layernorm_bwd_g_workspace(int rows, int d, int rows_per_group);
int mtl_layernorm_bwd_g(void* stream, const float* dy, const float* xhat, const float* rstd, const float* gamma,
                        const int* keep, const unsigned char* xmask, float xscale, float* dz, float* dzm, float* dz2, float* dgamma,
                        float* dbeta, float* dsum, float* workspace, int rows, int d, int defer_reduce, int rows_per_group,
                        long sParam, long sGrad);
/* the deferred parameter reductions of several mtl_layernorm_bwd calls in ONE launch (table in device memory; nw = workspace bytes
 * / (3 * d * 4); dmax = largest d of the table) */
typedef struct mtl_ln_reduce_desc {
    const float* part;
    float* dgamma;
    float* dbeta;
    float* dsum; /* nullable */
    int nw, d;
} mtl_ln_reduce_desc;
int mtl_ln_param_reduce_batch(void* stream, const mtl_ln_reduce_desc* table_dev, int n, int dmax);

/* ---- masked softmax: modules/common_layers.py:322-327.  S is [B][H][Tq][ld], in place.
 * keys k >= klen[b] (klen nullable) and, if causal, k > q are filled with -inf before the softmax. */
/* pmask (nullable u8 keep-mask, same [B][H][Tq][ld] layout) / pscale: dropout on the probabilities (common_layers.py:328);
 * P stays un-dropped in S (needed by the backward), the dropped copy that feeds P.V is written to P_dropped. */
int mtl_softmax_mask_fwd(void* stream, float* S, const int* klen, int causal, float scale, int B, int H, int Tq, int Tk,
                         int ld, const unsigned char* pmask, float pscale, float* P_dropped);
int mtl_softmax_bwd(void* stream, const float* P, float* dP /*in place -> dS*/, float scale, long rows, int Tk, int ld,
                    const unsigned char* pmask, float pscale);

/* ---- fused scaled-dot-product attention: modules/common_layers.py:317-331 (bmm, /temperature, masked_fill(-inf), softmax,
 * dropout, bmm) with the head split / merge of :291-293,301 done by stride and the masks of :296 derived in-kernel.
 * q (B*Tq rows, row stride ldq floats), k, v (B*Tk rows): head h occupies columns [h*dk, (h+1)*dk) of a row (dv for v / O).
 * keys k >= klen[b] (klen nullable) and, if causal, k > q are masked.  O (B*Tq rows, ldo) = softmax(q.k^T * scale) [* pmask *
 * pscale] . v ; lse[(b*H+h)*Tq + q] = log-sum-exp of the scaled scores (saved for the backward).  The (B,H,Tq,Tk) score tensor
 * is never written: 64-query workgroups stream 64-key tiles with an online softmax; the backward recomputes the probabilities.
 * pmask (nullable): u8 keep-mask [B][H][Tq][ldm] from mtl_dropout_mask, pscale = 1/(1-p).  (dk, dv) in {(64,64), (16,16)}
 * (mtl_attn_supported); q/k/v/dO 16-byte aligned with row strides that are multiples of 4.
 * Backward: dq, dk, dv are OVERWRITTEN (head-strided like their inputs); delta: B*H*Tq floats of scratch. */
int mtl_attn_supported(int dk, int dv);
int mtl_attn_fwd(void* stream, const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int* klen,
                 int causal, float scale, int B, int H, int Tq, int Tk, int dk, int dv, const unsigned char* pmask, int ldm,
                 float pscale, float* O, int ldo, float* lse);
int mtl_attn_bwd(void* stream, const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int* klen,
                 int causal, float scale, int B, int H, int Tq, int Tk, int dk, int dv, const unsigned char* pmask, int ldm,
                 float pscale, const float* O, const float* dO, int ldo, const float* lse, float* delta, float* dq, float* dk_,
                 float* dv_, int lddq, int lddk, int lddv);

/* ---- embedding + positional encoding: modules/decoder.py:96 ------------------------------------------- */
int mtl_embed_pe_fwd(void* stream, const long* ids, const float* table, const float* pe, float* out, int rows, int T, int d,
                     const unsigned char* mask /*nullable: dropout keep-mask*/, float mscale);
/* first[r] = 1 if no earlier row has ids[r]; next[r] = next row with the same id or -1 (both host-built): duplicates are
 * summed in row order by the thread of the chain head, so the scatter-add is one launch, parallel AND deterministic. */
int mtl_embed_bwd(void* stream, const long* ids, const int* first, const int* next, const float* dout,
                  float* dtable /*accum*/, int rows, int d, long pad_id, const unsigned char* mask, float mscale);

/* Task-grouped forms: rows in groups of rows_per_group; group g uses table + g * sParam / dtable + g * sGrad.  The occurrence
 * chains (first / next) must not cross groups. */
int mtl_embed_pe_fwd_g(void* stream, const long* ids, const float* table, const float* pe, float* out, int rows, int T, int d,
                       const unsigned char* mask, float mscale, int rows_per_group, long sParam);
int mtl_embed_bwd_g(void* stream, const long* ids, const int* first, const int* next, const float* dout, float* dtable, int rows,
                    int d, long pad_id, const unsigned char* mask, float mscale, int rows_per_group, long sGrad);

/* ---- dropout keep-masks: keep[i] = 1 with probability 1-p, Philox4x32-10(counter = offset + i/4, key = *seed_dev).
 * Active sites in the meta loop (model.train()): modules/decoder.py:96, modules/common_layers.py:130,303,328. */
int mtl_dropout_mask(void* stream, unsigned char* keep, long n, float p, const long* seed_dev, unsigned long long offset);

/* ---- cross-entropy + arg-max: utils/metrics.py:113-126, models/asr/transformer.py:146-147 ----------------
 * loss_out[0] = sum_rows(gold!=pad ? -log softmax(logits)[gold] : 0) / n_nonpad ; hyp = lowest arg-max index. */
/* inv_count_dev (nullable): device scalar 1/n_nonpad used instead of n_nonpad (keeps a captured hipGraph batch-independent) */
int mtl_ce_argmax_fwd(void* stream, const float* logits, const long* gold, int rows, int V, int ld, long pad_id,
                      float smoothing, int n_nonpad, const float* inv_count_dev, float* lse, long* hyp, float* rowloss,
                      float* loss_out);
/* dlogits = gscale * (gscale_dev ? *gscale_dev : 1) * (softmax - target) on non-pad rows, 0 elsewhere */
int mtl_ce_bwd(void* stream, const float* logits, const float* lse, const long* gold, int rows, int V, int ld, long pad_id,
               float smoothing, float gscale, const float* gscale_dev, float* dlogits, int ldd);

/* Task-grouped forms (rows % rows_per_group == 0): loss_out[g] = sum over the rows of group g * inv_count_dev[g] (one loss per
 * task, each normalised by ITS non-pad token count); the backward scales group g by gscale * gscale_dev[g]. */
int mtl_ce_argmax_fwd_g(void* stream, const float* logits, const long* gold, int rows, int V, int ld, long pad_id, float smoothing,
                        const float* inv_count_dev, float* lse, long* hyp, float* rowloss, float* loss_out, int rows_per_group);
int mtl_ce_bwd_g(void* stream, const float* logits, const float* lse, const long* gold, int rows, int V, int ld, long pad_id,
                 float smoothing, float gscale, const float* gscale_dev, float* dlogits, int ldd, int rows_per_group);

/* ---- out[c] += sum_r X[r*ld + c]  (bias gradients) -------------------------------------------------------- */
long mtl_colsum_workspace(long rows, int cols);
/* amax (optional, MTL_AMAX_FLOATS floats): all slot heads set to max|X| (written, not accumulated) -- the same pass over X */
int mtl_colsum_accum(void* stream, const float* X, long rows, int cols, long ld, float* out, float* workspace, float* amax);

/* ---- flat-parameter updates over ONE contiguous fp32 buffer (190 tensors in the reference) ----------------
 * inner SGD  trainer/asr/transient_trainer.py:106,207 -> theta1 = theta0 - alpha*g (theta0 is never mutated, which
 *            replaces deepcopy(state_dict)/load_state_dict at :155-160,237)
 * copy_grad  models/asr/transformer.py:205-240       -> mtl_axpy(G, g, 1)
 * clip       transient_trainer.py:205-206,253-254     -> mtl_sumsq(mode 2) + mtl_scale(a_dev)
 * Adam       transient_trainer.py:109,255 (torch defaults) */
int mtl_sgd_theta_prime(void* stream, const float* theta0, const float* g, float alpha, float* theta1, long n);
/* the inner steps of all local tasks in one launch: theta1[t*n + i] = theta0[i] - alpha * g[t*n + i], t < tasks (n % 4 == 0) */
int mtl_sgd_theta_prime_tasks(void* stream, const float* theta0, const float* g, float alpha, float* theta1, long n, int tasks);
/* out[i] (+)= sum_t x[t*n + i] in task order (copy_grad accumulation of a task stack: models/asr/transformer.py:219-229) */
int mtl_sum_tasks(void* stream, float* out, const float* x, long n, int tasks, int accumulate);
/* the same over a SLICE of the stack: out[i] (+)= sum_t x[t*task_stride + i], i < n (n, task_stride % 4 == 0, 16-byte aligned): one
 * parameter group (decoder | encoder | conv) of copy_grad as soon as its gradients are final, so that its all-reduce can start under
 * the rest of the backward (trainer/asr/transient_trainer.py:229,247-255) */
int mtl_sum_tasks_strided(void* stream, float* out, const float* x, long n, int tasks, long task_stride, int accumulate);
int mtl_axpy(void* stream, float* y, const float* x, float a, long n);
int mtl_copy_f32(void* stream, float* dst, const float* src, long n); /* device-to-device, asynchronous on `stream` */
int mtl_scale(void* stream, float* y, float a, const float* a_dev /*nullable: overrides a*/, long n);
int mtl_adam_step(void* stream, float* theta, const float* G, float* m, float* v, int step, float lr, float beta1,
                  float beta2, float eps, long n);
int mtl_sumsq(void* stream, const float* x, long n, float* out, float* workspace /* >= 4 KiB */, int mode, float arg);

/* ---- spectrogram front-end (utils/data_loader.py:65-96): after the STFT has been computed as
 * mtl_gemm_f32(frames (T x n_fft, lda = hop: overlapping rows of the padded waveform) . windowed DFT basis (n_fft x 2F)),
 * out[f*T + t] = log1p(|X[t][f]|), optionally followed by (x - mean) / std (unbiased) over the whole utterance.
 * partials: >= 256 doubles of scratch. */
int mtl_spect_logmag(void* stream, const float* reim, int ld, int T, int F, float* out, double* partials, int normalize);

/* ---- LSTM cell, one time step (lm/model/rnn_model.py:20 nn.LSTM; lm/main_meta_transfer.py:277-411) ----------
 * gx = x_t . W_ih^T + b_ih and gh = h_{t-1} . W_hh^T + b_hh come from mtl_gemm_f32_ex (B x 4H each, torch gate order i|f|g|o).
 * forward: acts = activated gates (saved), c = f*c_prev + i*g, h = o*tanh(c); h_drop (nullable) = h [* mask * mscale]: the
 * copy that feeds the next layer (nn.LSTM's inter-layer dropout) / the decoder (self.drop).
 * backward: dh = dh_up [* mask * mscale] + dh_rec (both nullable), dc = dc_next (nullable) + dh*o*(1-tanh(c)^2);
 * dgates = pre-activation gradients (B x 4H), dc_prev = dc*f. */
int mtl_lstm_cell_fwd(void* stream, const float* gx, const float* gh, const float* c_prev, float* acts, float* c, float* h,
                      float* h_drop, const unsigned char* mask, float mscale, int B, int H);
int mtl_lstm_cell_bwd(void* stream, const float* dh_up, const unsigned char* mask, float mscale, const float* dh_rec,
                      const float* dc_next, const float* acts, const float* c, const float* c_prev, float* dgates, float* dc_prev,
                      int B, int H);
/* Persistent LSTM layer (csrc/mtl_lstm.hip; lm/model/rnn_model.py:20 nn.LSTM over a bptt window): ALL T time steps of one layer in one
 * launch per direction.  Workgroup w owns hidden units [8 w, 8 w + 8) for the whole sequence (its 32 gate rows of W_hh stay in
 * registers), one grid-wide hand-off per step (write-through payload + arrival counter, bounded spins; the grid is H / 8 <= 64
 * workgroups).  The backward splits dh_rec = dG_{t+1} W_hh by rows: every workgroup publishes the partial of its own 32 columns of dG
 * and sums the partials of its 8 units in a fixed order (two partial buffers in the workspace).
 *   forward : gx (T B, 4H) = x W_ih^T + b_ih of all steps (one product), hall / call (T + 1, B, H) with slot 0 = the incoming state;
 *             writes hall[1..T], call[1..T], acts (T B, 4H) [i|f|g|o after the nonlinearities], xout (T B, H) = h (dropout applied when
 *             mask != NULL: u8 keep flags, scale mscale) -- exactly what T calls of the recurrent product + mtl_lstm_cell_fwd produce.
 *   backward: dx_up (T B, H) gradient of xout (may be NULL), writes dG (T B, 4H) = gradient of the gate pre-activations of every step
 *             (truncated BPTT: no gradient into the incoming state) -- what T calls of mtl_lstm_cell_bwd + the dh_rec product produce.
 * workspace: mtl_lstm_layer_workspace() bytes of device memory, 256-byte aligned (4 KB header: arrival counter, per-workgroup step flags and the
 * error word -- u32 [1] != 0 after a timed-out wait; sticky: launches do not clear it, the caller zeroes the workspace once and checks the
 * word whenever it synchronises anyway -- followed by the backward's partial buffers).
 * mtl_lstm_layer_supported: 1 <= B <= 32 and H in {128, 256, 384, 512}; other shapes take the per-step calls. */
int mtl_lstm_layer_supported(int B, int H);
long mtl_lstm_layer_workspace(void);
int mtl_lstm_layer_fwd(void* stream, const float* gx, const float* w_hh, const float* b_hh, float* hall, float* call, float* acts,
                       float* xout, const unsigned char* mask, float mscale, int T, int B, int H, void* workspace);
int mtl_lstm_layer_bwd(void* stream, const float* dx_up, const unsigned char* mask, float mscale, const float* w_hh, const float* acts,
                       const float* call, float* dG, int T, int B, int H, void* workspace);

/* The whole layer stack as ONE wavefront launch per direction (csrc/mtl_lstm.hip; lm/model/rnn_model.py:20 nn.LSTM(ninp, nhid, nlayers,
 * dropout=...)): layer l runs one step behind layer l - 1 instead of after it.  `layers` is a HOST struct of device pointers, one
 * entry per layer (read during the call; not recordable in a command list): gx[0] holds layer 0's input contributions
 * x W_ih0^T + b_ih0 over all T steps (w_ih[0] / b_ih[0] are not used), gx[l >= 1] is T B x 4H scratch the call fills; layers >= 1
 * must have input width H; hall/call/acts/xout/mask/dG as in the single-layer calls (mask[l]: the dropout on layer l's output, or
 * NULL).
 *   forward : fills hall, call, acts, xout of every layer.
 *   backward: fills dG of every layer from dx_up (gradient w.r.t. the top layer's dropped output); scratch = mtl_lstm_stack_scratch()
 *             bytes (the per-step partial sums handed from a layer to the one below).
 * Results equal the per-layer calls up to the summation order of the input contribution (formed per step instead of by one product).
 * mtl_lstm_stack_supported: the single-layer limits, NL <= 4 and (2 NL - 1) * H / 8 <= 224 workgroups (all resident). */
#define MTL_LSTM_MAX_LAYERS 4
typedef struct mtl_lstm_stack {
    const float *w_ih[MTL_LSTM_MAX_LAYERS], *b_ih[MTL_LSTM_MAX_LAYERS], *w_hh[MTL_LSTM_MAX_LAYERS], *b_hh[MTL_LSTM_MAX_LAYERS];
    float *gx[MTL_LSTM_MAX_LAYERS], *hall[MTL_LSTM_MAX_LAYERS], *call[MTL_LSTM_MAX_LAYERS], *acts[MTL_LSTM_MAX_LAYERS], *xout[MTL_LSTM_MAX_LAYERS], *dG[MTL_LSTM_MAX_LAYERS];
    const unsigned char* mask[MTL_LSTM_MAX_LAYERS];
} mtl_lstm_stack;
int mtl_lstm_stack_supported(int B, int H, int NL);
long mtl_lstm_stack_scratch(int T, int B, int H, int NL);
int mtl_lstm_stack_fwd(void* stream, const mtl_lstm_stack* layers, float mscale, int T, int B, int H, int NL, void* workspace);
int mtl_lstm_stack_bwd(void* stream, const mtl_lstm_stack* layers, const float* dx_up, float mscale, float* scratch, int T, int B, int H,
                       int NL, void* workspace);


/* ---- raw byte helpers on a stream (so that a whole task body consists of library calls only and can be replayed) ---- */
int mtl_memset_zero(void* stream, void* dst, long bytes);
int mtl_memcpy_d2d(void* stream, void* dst, const void* src, long bytes);
/* hipEventRecord / hipStreamWaitEvent on caller-owned handles (fork / join of the parameter-gradient side stream) */
int mtl_event_record(void* event, void* stream);
int mtl_stream_wait_event(void* stream, void* event);

/* ---- command lists: ONE call from the host language replays a recorded sequence of the calls above ------------------------
 * The per-task body of the meta step (trainer/asr/transient_trainer.py:178-237: train pass, inner SGD, validation pass,
 * copy_grad accumulation) is ~700 library calls whose arguments -- device pointers into the caller's static buffers, shapes,
 * scalars, stream and event handles -- do not change from task to task: everything batch-dependent lives in device buffers the
 * host refreshes before the replay.  The host layer records the calls of one eager run (`mtl_cmd` entries: opcode = position of
 * the function in this header's declaration order, see mtl_cmdlist_opcode; arguments in declaration order, one 8-byte slot
 * each: pointers / integers as-is, floats as double) and afterwards issues mtl_cmdlist_run, which performs exactly those calls
 * in order from C.  Returns 0, or the first failing call's code with *failed_index = its position (nullable). */
typedef struct mtl_cmd {
    int op;
    int nargs;
    union {
        void* p;
        long l;
        double d;
    } a[48];
} mtl_cmd;
int mtl_cmdlist_opcode(const char* function_name);     /* -1 if the function cannot be recorded */
int mtl_cmdlist_run(const mtl_cmd* cmds, int n, int* failed_index);
/* diagnostics: the same replay, and host_us[i] = HOST time (microseconds) the i-th call took to return (a launch that blocks in
 * the runtime -- full queue, exhausted kernel-argument pool, a signal still in use -- shows up here, not in any device profile) */
int mtl_cmdlist_run_timed(const mtl_cmd* cmds, int n, int* failed_index, float* host_us);

/* ---- host helper: Levenshtein distance on code points (utils/metrics.py:38-44 uses python-Levenshtein) ---- */
int mtl_levenshtein_u32(const unsigned int* a_host, int na, const unsigned int* b_host, int nb);

#ifdef __cplusplus
}
#endif
#endif /* MTL_HIP_H */

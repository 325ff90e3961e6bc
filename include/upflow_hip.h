/*
 * upflow_hip.h — C-ABI of libupflow_hip.so: the MI355X (gfx950) native operators of the UPFlow
 * hot path.  Plain pointers, sizes and a HIP stream; no torch types.
 *
 * This is the drop-in boundary.  The reference crosses Python -> native exactly once on this path,
 * through the pybind11 module `correlation_cuda`
 *     /root/reference/model/correlation_package/correlation_cuda.cc:169-172
 *         forward (input1,input2,rInput1,rInput2,output, pad,k,max_disp,s1,s2,mult)      :10-87
 *         backward(input1,input2,rInput1,rInput2,gradOutput,gradInput1,gradInput2, ...)  :89-167
 * and reaches ATen's native grid_sample / interpolate kernels at
 *     model/pwc_modules.py:79,200,205   utils/tools.py:1304   model/upflow.py:79-88.
 * Each entry point below names the reference interface it replaces.  The reference-side binding
 * a maintainer would add is in INTEGRATION.md; ours is upflow_pytorch_amd/_lib.py (ctypes).
 *
 * Conventions (all entry points)
 *   - device pointers, tensors are contiguous NCHW (the reference kernels assume that too,
 *     correlation_cuda_kernel.cu:15-39 ignores the strides it is handed);
 *   - `dtype` selects the element type of feature-like tensors: UPF_F32 / UPF_F16 / UPF_BF16;
 *     flows, sampling positions, masks and every accumulator are always fp32;
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); calls only enqueue work, they
 *     never synchronise, allocate or free, so they can be captured into a hipGraph;
 *   - return value: 0 (UPF_OK) on success, a negative UPF_E* code on a rejected argument, a
 *     positive hipError_t if the launch failed; upf_last_error() gives the message of the last
 *     failure on the calling thread (the reference printf()s and returns 0,
 *     correlation_cuda_kernel.cu:383-392, which its .cc turns into AT_ERROR, :81-83);
 *   - threading: like the reference's FFI (called with the GIL held, one process per GPU) the library expects
 *     one calling thread per device at a time; its only process-wide state is launch bookkeeping (kernel
 *     attributes set on first use, the heuristics of upf_conv_set_option) — no allocations, no caches of data.
 */
#ifndef UPFLOW_HIP_H
#define UPFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { UPF_F32 = 0, UPF_F16 = 1, UPF_BF16 = 2 };
enum { UPF_OK = 0, UPF_EINVAL = -1, UPF_EDTYPE = -2, UPF_EALIGN = -3, UPF_EUNSUPPORTED = -4 };
enum { UPF_MASK_NONE = 0, UPF_MASK_LITERAL = 1, UPF_MASK_ROBUST = 2 };

/* library identity: "upflow_hip <version> gfx950" */
const char* upf_version(void);
const char* upf_last_error(void);

/* ---- cost volume ------------------------------------------------------------------------------
 * 81-neighbour correlation, (pad,k,max_disp,s1,s2) = (4,1,4,1,1), the only configuration the model
 * instantiates (model/upflow.py:561-562):
 *     out[n, 9*(dy+4)+(dx+4), y, x] = (1/C) * sum_c f1[n,c,y,x] * f2[n,c,y+dy,x+dx]   (f2 = 0 outside)
 * Replaces correlation_cuda.forward (correlation_cuda.cc:10-87 -> correlation_cuda_kernel.cu:302-393:
 * 2x channels_first + correlation_forward) without the padded-NHWC staging copies.
 *   f1,f2 : [B,C,H,W] of `dtype`;  out : [B,81,H,W] of `dtype`, batch stride `out_batch_stride`
 *   elements (0 = 81*H*W) so the volume can be written straight into the first 81 channels of the
 *   115-channel estimator input (model/upflow.py:565).
 *   leaky_slope != 0 fuses LeakyReLU(slope) (model/upflow.py:563-564); 0 = plain correlation.
 * fp32 accumulation for every dtype.
 */
int upf_corr81_forward(const void* f1, const void* f2, void* out,
                       int B, int C, int H, int W, int dtype,
                       long long out_batch_stride, float leaky_slope, void* stream);

/* Measurement helper (bench.py's `roofline` object): `nrep` back-to-back launches of the kernel above,
 * each bracketed by its own pair of HIP events recorded on `stream` right around the kernel
 * (hipExtLaunchKernel start/stop events); synchronises the stream and returns the average and the
 * minimum kernel duration in microseconds.  Not capturable (it synchronises). */
int upf_corr81_forward_timed(const void* f1, const void* f2, void* out,
                             int B, int C, int H, int W, int dtype,
                             long long out_batch_stride, float leaky_slope, void* stream,
                             int nrep, float* avg_us, float* min_us);

/* Feature normalisation fused into the cost volume (model/upflow.py:549-562):
 *     out = corr81(normalize(f1), normalize(f2))        normalize = upf_normalize_forward's arithmetic, per tensor
 * in TWO launches — one statistics pass over both tensors, then the cost volume whose loader normalises every element
 * (and re-rounds it to `dtype`) on its way into LDS — instead of four, and without writing / re-reading the two
 * normalised feature maps.  Bit-identical to upf_normalize_forward x2 + upf_corr81_forward.  bf16 / fp16, C <= 208
 * (upf_corr81_norm_supported), W >= 4; `workspace`: upf_corr81_norm_workspace_bytes(B,C,H,W) bytes of device memory. */
int upf_corr81_norm_supported(int C, int dtype);
long long upf_corr81_norm_workspace_bytes(int B, int C, int H, int W);
int upf_corr81_norm_forward(const void* f1, const void* f2, void* out,
                            int B, int C, int H, int W, int dtype,
                            long long out_batch_stride, float leaky_slope, void* workspace, void* stream);
/* The same into channel octets (round 3; the flow estimator's input buffer in the C8 layout of upf_conv_forward_c8):
 * out8 = the first of 11 octets of a [n][octet][H][W][8] buffer, out8_batch_stride its batch stride in elements.
 * Octet j < 9: displacements (dy = j-4, dx = -4..+3) in positions 0..7; octet 9 position p: (dy = p-4, dx = +4);
 * octet 10 position 0: (+4,+4), positions 1..7 zero.  The convolution reading it learns this order through the k-map
 * of upf_conv_pack_weights_kmap.  Values are bit-identical to upf_corr81_norm_forward's.  W % 8 == 0. */
int upf_corr81_norm_forward_c8(const void* f1, const void* f2, void* out8, long long out8_batch_stride,
                               int B, int C, int H, int W, int dtype, float leaky_slope, void* workspace, void* stream);
/* PITCHED feature rows (round 5): the H rows of every channel plane of f1 / f2 are f_row_pitch elements apart (>= W; plane stride
 * H * f_row_pitch, item stride C * H * f_row_pitch; 0 = W, i.e. the contiguous forms above).  With a pitch that is a multiple of 8
 * every row is 16-byte aligned whatever W is, so RAGGED pyramid levels — KITTI's native 375x1242 frames: W = 311, 156, 78, 39, 20,
 * the frame size the reference's evaluation feeds (test.py:40-47, dataset/kitti_dataset.py:609-631) — keep the aligned loads and,
 * for the octet form, any W >= 4 is accepted (one 16-byte entry per pixel: the output is aligned for every W).  Nothing depends on
 * what the pitch padding holds (the loader masks it; the statistics skip it); results are bit-identical to the contiguous forms. */
int upf_corr81_norm_forward_pitched(const void* f1, const void* f2, int f_row_pitch, void* out,
                                    int B, int C, int H, int W, int dtype,
                                    long long out_batch_stride, float leaky_slope, void* workspace, void* stream);
int upf_corr81_norm_forward_c8_pitched(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride,
                                       int B, int C, int H, int W, int dtype, float leaky_slope, void* workspace, void* stream);
/* MIXED storage types (round 5, the `pyramid_dtype` option: fp16 pyramid features, bf16 decoder buffers): the features are `dtype`,
 * the cost volume is rounded ONCE from its fp32 sums to `out_dtype` (the other 16-bit type, or the same: the entry points above). */
int upf_corr81_norm_forward_mixed(const void* f1, const void* f2, int f_row_pitch, void* out,
                                  int B, int C, int H, int W, int dtype, int out_dtype,
                                  long long out_batch_stride, float leaky_slope, void* workspace, void* stream);
int upf_corr81_norm_forward_c8_mixed(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride,
                                     int B, int C, int H, int W, int dtype, int out_dtype, float leaky_slope, void* workspace, void* stream);
/* measurement helper (bench.py): one statistics launch, then nrep launches of the normalising cost volume, each between
 * its own pair of HIP events on `stream` — the kernel that runs inside the inference step */
int upf_corr81_norm_forward_timed(const void* f1, const void* f2, void* out,
                                  int B, int C, int H, int W, int dtype,
                                  long long out_batch_stride, float leaky_slope, void* workspace, void* stream,
                                  int nrep, float* avg_us, float* min_us);
/* ... and of upf_corr81_norm_forward_c8 (the form the step launches at the levels whose estimator runs on octets) */
int upf_corr81_norm_forward_c8_timed(const void* f1, const void* f2, void* out8, long long out8_batch_stride,
                                     int B, int C, int H, int W, int dtype, float leaky_slope, void* workspace, void* stream,
                                     int nrep, float* avg_us, float* min_us);

int upf_corr81_norm_forward_c8_timed_pitched(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride,
                                             int B, int C, int H, int W, int dtype, float leaky_slope, void* workspace, void* stream,
                                             int nrep, float* avg_us, float* min_us);

/* ... with the output's storage type given separately (upf_corr81_norm_forward_c8_mixed: fp16 features -> bf16 octets, what the
 * bf16 step launches since its feature pyramid stays fp16 — UPFlow_net.to_inference, round 6) */
int upf_corr81_norm_forward_c8_timed_mixed(const void* f1, const void* f2, int f_row_pitch, void* out8, long long out8_batch_stride,
                                           int B, int C, int H, int W, int dtype, int out_dtype, float leaky_slope, void* workspace, void* stream,
                                           int nrep, float* avg_us, float* min_us);

/* Launch heuristics of the 16-bit cost volume, for tuning and for the tests to reach every kernel variant:
 *   "variant"  (-1)  -1 = choose by shape; 0..3 = force tile geometry 8x32 / 4x32 / 2x32 / 4x16 where C fits
 *   "old_path" (0)   1 = the channel-chunked kernels (corr81_mfma_kernel / corr81_fwd_kernel) instead of the
 *                    all-channels-in-LDS kernel
 * returns the previous value, or -1000 for an unknown name. */
int upf_corr_set_option(const char* name, int value);

/* Gradients of the above (correlation_cuda.backward, correlation_cuda.cc:89-167 ->
 * correlation_cuda_kernel.cu:116-300, 396-530):
 *   g1[n,c,y,x] = (1/C) sum_d gO[n,d,y,x]       * f2[n,c,y+dy,x+dx]
 *   g2[n,c,y,x] = (1/C) sum_d gO[n,d,y-dy,x-dx] * f1[n,c,y-dy,x-dx]
 * grad_out : [B,81,H,W] of `dtype` (contiguous);  g1,g2 : [B,C,H,W] of `dtype`. */
int upf_corr81_backward(const void* f1, const void* f2, const void* grad_out, void* g1, void* g2,
                        int B, int C, int H, int W, int dtype, void* stream);

/* General-parameter cost volume with the reference's full argument list (correlation_cuda.cc:10-17):
 * any pad/kernel/max_displacement/stride1/stride2; output [B,(2*(md/s2)+1)^2,outH,outW] with
 * outH = ceil((H+2*pad-2*((k-1)/2+md))/s1) (correlation_cuda.cc:24-34).  Dispatches to the tuned
 * 81-neighbour kernel for (4,1,4,1,1), otherwise to a plain one-thread-per-output kernel.
 * corr_type_multiply is accepted and unused, as in the reference (correlation_cuda_kernel.cu:302+). */
int upf_correlation_forward(const void* in1, const void* in2, void* out,
                            int B, int C, int H, int W, int dtype,
                            int pad_size, int kernel_size, int max_displacement,
                            int stride1, int stride2, int corr_type_multiply, void* stream);
/* Gradients of upf_correlation_forward (correlation_cuda.backward with the full argument list, correlation_cuda.cc:89-167 ->
 * correlation_cuda_kernel.cu:116-300): the tuned kernels for (4,1,4,1,1); a plain one-thread-per-element kernel for any
 * (pad, md, stride2) with kernel_size 1 and stride1 1 — the parameter sets for which the reference's backward kernels are the
 * gradient of its forward (their block -> pixel map and (ymin, ymax) windows, :129-141, :222-256, are not for stride1 > 1 or
 * kernel_size > 1); UPF_EUNSUPPORTED otherwise.  grad_out [B,(2*(md/s2)+1)^2,outH,outW], g1 / g2 [B,C,H,W], all of `dtype`.
 * upf_correlation_forward itself returns UPF_EUNSUPPORTED where the reference reads outside its padded buffers
 * (kernel_size > 1 with md - (md/s2)*s2 < (k-1)/2: correlation_cuda_kernel.cu:62, :87-91). */
int upf_correlation_backward(const void* in1, const void* in2, const void* grad_out, void* g1, void* g2,
                             int B, int C, int H, int W, int dtype,
                             int pad_size, int kernel_size, int max_displacement,
                             int stride1, int stride2, int corr_type_multiply, void* stream);
/* output geometry of upf_correlation_forward */
int upf_correlation_out_shape(int H, int W, int pad_size, int kernel_size, int max_displacement,
                              int stride1, int stride2, int* out_channels, int* out_h, int* out_w);

/* ---- backward warp ----------------------------------------------------------------------------
 * y[n,c,i,j] = bilinear sample of x[n,c] at (j + flow[n,0,i,j], i + flow[n,1,i,j]), zeros outside,
 * torch-1.1 grid_sample semantics (align_corners=True), times a validity mask:
 *   UPF_MASK_NONE    tools.torch_warp                  utils/tools.py:1274-1319
 *   UPF_MASK_LITERAL WarpingLayer_no_div.forward       model/pwc_modules.py:184-207
 *                    mask = grid_sample(ones) >= 1.0, reproduced bit-exactly: the four weight*tap
 *                    products in fp32, summed ((nw+ne)+sw)+se, no FMA contraction
 *   UPF_MASK_ROBUST  exact in-bounds predicate (non-default; SURVEY.md §7-H2 protocol P3b)
 *   x,y : [B,C,H,W] of `dtype`;  flow : [B,2,H,W] fp32.
 *   batch_shift: output item n samples x[(n + batch_shift) % B] — with both frames of a pair stacked along
 *   the batch ([im1 features; im2 features]) batch_shift = B/2 warps "the other frame" for both flow
 *   directions in one launch, no gather copy.  0 = plain. */
int upf_warp_forward(const void* x, const float* flow, void* y,
                     int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream);
/* same, x / y being channel slices of wider contiguous NCHW buffers (batch strides in elements, 0 = C*H*W): the
 * inference path warps straight out of / into the concatenation buffers the convolutions read (no slot copies). */
int upf_warp_forward_strided(const void* x, long long x_batch_stride, const float* flow, void* y, long long y_batch_stride,
                             int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream);
/* same with PITCHED rows (round 5): rows of x / y are x_row_pitch / y_row_pitch elements apart (>= W, 0 = W; plane stride H * pitch,
 * batch strides 0 = C * H * pitch); the flow stays a contiguous fp32 tensor.  With an even y pitch odd-width levels keep the 4-byte
 * (two-pixel) stores — the pixel beyond W lands in the row's own padding.  Same values as the contiguous forms. */
int upf_warp_forward_pitched(const void* x, long long x_batch_stride, int x_row_pitch, const float* flow, void* y, long long y_batch_stride,
                             int y_row_pitch, int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream);
/* same on CHANNEL-OCTET tensors ("C8": [n][C/8][H][W][8], see upf_conv_forward_c8; bf16 / fp16): x8 / y8 point at the first
 * octet plane of octet slices of C8 buffers (batch strides in elements), n_oct octets.  Values equal the NCHW kernel's. */
int upf_warp_forward_c8(const void* x8, long long x_batch_stride, const float* flow, void* y8, long long y_batch_stride,
                        int B, int n_oct, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream);
/* Gradients of the warp: gx [B,C,H,W] of `dtype` (wrt x; with batch_shift it lands on the item that was sampled) and
 * gflow [B,2,H,W] fp32.  The scatter into gx is accumulated in 64-bit fixed point (value * 2^44, integer atomics), so the
 * result is bit-reproducible run to run; `workspace`: upf_warp_backward_workspace_bytes(B,C,H,W) bytes of device memory
 * (zero-filled by the call itself).  grad_y : [B,C,H,W] of `dtype`. */
long long upf_warp_backward_workspace_bytes(int B, int C, int H, int W);
int upf_warp_backward(const void* x, const float* flow, const void* grad_y,
                      void* gx, float* gflow, void* workspace,
                      int B, int C, int H, int W, int dtype, int mask_mode, int batch_shift, void* stream);

/* ---- flow up-sampling -------------------------------------------------------------------------
 * upsample2d_flow_as / upsample_flow (model/pwc_modules.py:77-104): bilinear align_corners=True
 * resize [B,C,h,w] -> [B,C,H,W] fp32; with if_rate channel 0 *= W/w and channel 1 *= H/h. */
int upf_flow_upsample_forward(const float* x, float* y, int B, int C, int h, int w, int H, int W,
                              int if_rate, void* stream);
int upf_flow_upsample_backward(const float* grad_y, float* gx, int B, int C, int h, int w, int H, int W,
                               int if_rate, void* stream);

/* ---- SGU interpolation-blend  (model/upflow.py:79-88) -----------------------------------------
 * x_out [B,3,h,w] of `dtype` = (inter_flow_x, inter_flow_y, mask_logit) from the SGU estimator.
 *   inter_flow = x_out[:, :2]; inter_mask = sigmoid(x_out[:, 2])
 *   decoder level (Hf = h, Wf = w):      flow_init [B,2,h,w]
 *   final level   (Hf,Wf = full size):   inter_flow, inter_mask bilinearly up-sampled (align_corners,
 *                                        flow channels * Wf/w, Hf/h), flow_init = output_level_flow
 *   flow_up = torch_warp(flow_init, inter_flow) * (1 - inter_mask) + flow_init * inter_mask
 * flow_init, flow_up : [B,2,Hf,Wf] fp32; inter_flow [B,2,Hf,Wf], inter_mask [B,1,Hf,Wf] fp32 are
 * optional outputs (NULL = do not materialise).  One launch instead of ~8 ATen launches.
 * At the final level the sigmoid of the mask logits is evaluated once per LOW-resolution pixel into `workspace`
 * (upf_sgu_blend_forward_workspace_bytes(B,h,w,Hf,Wf) bytes; 0 / NULL at a decoder level) and interpolated from there. */
long long upf_sgu_blend_forward_workspace_bytes(int B, int h, int w, int Hf, int Wf);
int upf_sgu_blend_forward(const float* flow_init, const void* x_out, float* flow_up,
                          float* inter_flow, float* inter_mask, void* workspace,
                          int B, int h, int w, int Hf, int Wf, int dtype, void* stream);
/* The decoder-level blend (x_out at the flow's own resolution) that ALSO stores the blended flow rounded to x_out's 16-bit type into the
 * flow estimator's input buffer — two NCHW planes [B,2,H,W] (batch stride given) or one octet entry per pixel [u, v, 0 x 6] — which
 * upf_flow_update / upf_flow_update_c8 wrote in a launch of their own (round 6; inference). */
int upf_sgu_blend_forward_flow16(const float* flow_init, const void* x_out, float* flow_up, void* flow16, long long flow16_batch_stride,
                                 int flow16_is_c8, int B, int H, int W, int dtype, void* stream);
/* g_flow_init32 [B,2,Hf,Wf] and g_x_out32 [B,3,h,w] fp32, fully produced by the call.  The scatters accumulate in 64-bit
 * fixed point (bit-reproducible); `workspace`: upf_sgu_blend_backward_workspace_bytes(B,h,w,Hf,Wf) bytes, zero-filled
 * by the call itself on `stream`. */
long long upf_sgu_blend_backward_workspace_bytes(int B, int h, int w, int Hf, int Wf);
int upf_sgu_blend_backward(const float* flow_init, const void* x_out, const float* grad_flow_up,
                           float* g_flow_init32, float* g_x_out32, void* workspace,
                           int B, int h, int w, int Hf, int Wf, int dtype, void* stream);

/* ---- feature normalisation  (network_tools.normalize_features, model/upflow.py:94-137) ---------
 * inference flags of test.py:22-30: per sample, per channel mean and UNBIASED variance over H*W,
 * y = (x - mean) / sqrt(var + 1e-16).  x,y : [N,HW] rows (N = B*C) of `dtype`;
 * mean, rstd : [N] fp32 optional outputs (saved for backward).  Rows are split over several workgroups
 * (two launches, deterministic merge): `workspace` must hold upf_normalize_workspace_bytes(N, HW) bytes. */
long long upf_normalize_workspace_bytes(long long N, int HW);
int upf_normalize_forward(const void* x, void* y, float* mean, float* rstd, void* workspace,
                          long long N, int HW, int dtype, void* stream);
int upf_normalize_backward(const void* y, const void* grad_y, const float* rstd, void* gx,
                           long long N, int HW, int dtype, void* stream);

/* ---- convolution on the matrix cores   -------------------------------------
 * The dense flow-estimator / context / SGU-estimator / pyramid convolutions (model/pwc_modules.py:122-142,
 * :250-286, :396-412; model/upflow.py:24-60, :349-353), bf16 / fp16 only (fp32 stays with MIOpen = the
 * parity mode), kernel_size 3 (dilation d <= 16, padding d; stride 1, or 2 with d = 1) or 1 (stride 1):
 *   y[n,co,i,j] = act(bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[n,ci, s*i+(ky-1)d, s*j+(kx-1)d])
 * output (H-1)/s+1 x (W-1)/s+1.  x / y point at the FIRST input / output channel of channel slices of larger
 * contiguous NCHW buffers (batch strides in elements), so the estimator's growing concatenation needs no
 * copies.  w_packed: upf_conv_pack_weights() output (MFMA lane order, zero padded to pad32(Cout) x pad32(Cin),
 * done once per layer).  Any Cout; any W >= 8 (rows that are not 16-byte aligned take a slightly slower
 * staging path); Cin*H*W*2 < 2^31.
 * upf_conv_set_option: launch heuristics, for tuning and for the tests to reach every kernel variant:
 *   "sk_grid"    (48)  grids of at most this many 8x32 pixel tiles use the split-K kernel (0 = never);
 *   "sk_grid_narrow" (96) the same bound for layers with Cout <= 64;  "sk_grid_d4" (16) for dilation 4
 *   "small_grid" (256) at most this many tiles: narrower workgroups, output channels over blockIdx.y
 *   "rpw4_min"   (256) at least this many 16x32 tiles: Cout <= 32 layers use 16-row tiles
 *   "ph_fit"     (1)   dilated layers: tile height (8 / 6 / 4 rows) fitted to the rows of a row phase (0 = always 8)
 *   "force_mtw"  (0)   experiments: 1 / 2 / 4 output-channel blocks per workgroup whatever the grid (0 = heuristic)
 *   "force_sk"   (-1)  experiments: 0 = never the split-K kernel, 1 = wherever it applies (-1 = heuristic)
 *   "ablate"     (0)   experiments: bit mask — 1 no matrix phase, 2 no x loads, 4 no LDS staging writes, 8 no weight loads
 * returns the previous value, or INT32_MIN for an unknown name. */
long long upf_conv_packed_bytes(int Cin, int Cout, int kernel_size);
int upf_conv_pack_weights(const void* w /* [Cout,Cin,k,k] */, void* w_packed, int Cin, int Cout, int kernel_size,
                          int dtype, void* stream);
int upf_conv_forward(const void* x, long long x_batch_stride, const void* w_packed, const float* bias,
                     void* y, long long y_batch_stride, int B, int Cin, int Cout, int H, int W,
                     int kernel_size, int dilation, int stride, float leaky_slope, int dtype, void* stream);
/* PITCHED rows (round 5): the rows of x / y are x_row_pitch / y_row_pitch elements apart (>= W / Wo; 0 = contiguous; plane stride
 * rows * pitch).  A pitch that is a multiple of 8 (with 16-byte aligned bases and batch strides) selects the aligned staging and the
 * 16-byte epilogue for ANY logical width: the 8-pixel group that straddles W is loaded in place and its trailing pixels replaced by
 * the zero padding, an output segment that straddles Wo is stored whole (its tail lands in the row's own padding).  Nothing depends
 * on what the padding holds; results are bit-identical to upf_conv_forward on contiguous copies. */
int upf_conv_forward_pitched(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                             void* y, long long y_batch_stride, int y_row_pitch, int B, int Cin, int Cout, int H, int W,
                             int kernel_size, int dilation, int stride, float leaky_slope, int dtype, void* stream);
/* 1x1 convolution, Cout <= 32, whose OUTPUT type is the other 16-bit type (`pyramid_dtype`: the projection of the fp16 pyramid features
 * into the bf16 estimator / SGU buffers): x, w_packed of `dtype`, fp32 sums rounded once to `out_dtype`; y = NCHW planes (pitched like
 * upf_conv_forward_pitched) or, y_is_c8, channel octets (x rows 16-byte aligned). */
int upf_conv1x1_forward_mixed(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                              void* y, long long y_batch_stride, int y_row_pitch, int y_is_c8, int B, int Cin, int Cout, int H, int W,
                              float leaky_slope, int dtype, int out_dtype, void* stream);
/* The 1x1 projection of a level's features, NCHW -> octets, stored into TWO octet buffers by one launch (round 6): it is the input of
 * both dense stacks of a level — the flow estimator's buffer and the SGU estimator's (model/upflow.py:546-553 with :71-75) — and
 * used to be computed twice.  Same arithmetic and bits as upf_conv_forward_c8 / upf_conv1x1_forward_mixed (out_dtype may equal dtype). */
int upf_conv1x1_forward_c8_dual(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                                void* y8_a, long long ya_batch_stride, void* y8_b, long long yb_batch_stride,
                                int B, int Cin, int Cout, int H, int W, float leaky_slope, int dtype, int out_dtype, void* stream);
int upf_conv_set_option(const char* name, int value);

/* ---- the same convolutions for the fp32 PARITY mode: split-precision products on the fp16 matrix cores  (round 4;
 * csrc/conv_x3.hip).  The reference's convolutions are fp32 (model/pwc_modules.py:122-142, :250-286, :396-412,
 * model/upflow.py:24-60; cuDNN there, MIOpen here until round 4).  x, y, bias and w are fp32; both operands are split into two
 * fp16 halves (a = fp16(a) + fp16(a - fp16(a)), 22-23 bits) on the way into the kernel and every product is formed as
 * a_hi*b_hi + a_lo*b_hi + a_hi*b_lo (nprod = 3; nprod = 11: the low-order products accumulate in their own registers and
 * join the sum once, a third of the rounding chain) with fp32 accumulation in the MFMA: fp32-class results at about a third of
 * the 16-bit rate.  The weights are scaled by a per-layer power of two (computed on the device at pack time, stored in the packed
 * operand's 1 KB header) so that their low halves are normal fp16 numbers.  Range: |x| < 65504.
 * 3x3 with dilation 1..16 at stride 1, 3x3 at stride 2, 1x1; any H, W >= 1 (element-wise bounds at ragged / unaligned rows);
 * x / y are channel slices of contiguous NCHW fp32 buffers (batch strides in elements).  w_packed: upf_conv_x3_packed_bytes
 * bytes, filled by upf_conv_x3_pack_weights from w [Cout,Cin,k,k] fp32. */
long long upf_conv_x3_packed_bytes(int Cin, int Cout, int kernel_size);
int upf_conv_x3_pack_weights(const float* w, void* w_packed, int Cin, int Cout, int kernel_size, void* stream);
int upf_conv_x3_forward(const float* x, long long x_batch_stride, const void* w_packed, const float* bias, float* y,
                        long long y_batch_stride, int B, int Cin, int Cout, int H, int W, int kernel_size, int dilation,
                        int stride, float leaky_slope, int nprod, void* stream);
/* launch heuristics of upf_conv_x3_forward, for tuning and for the tests to reach both kernels: "sk_max_tiles" (96) — the split-K
 * kernel (2 x 32-pixel tiles, the four waves of a workgroup split the input channels; stride 1, >= 64 input channels, 3 products)
 * is chosen where the batch has at most this many 8 x 32-pixel tiles (0: never; 1 << 30: wherever eligible).
 * Returns the previous value, INT32_MIN for an unknown name. */
int upf_conv_x3_set_option(const char* name, int value);
/* writes 2^-6 to out_device[0] if v_mfma_f32_32x32x16_f16 multiplies fp16 SUBNORMAL inputs un-flushed (what the low halves of
 * small operands rely on), 0 if it flushes them */
int upf_mfma_f16_denorm_probe(float* out_device, void* stream);

/* ---- the same convolutions with operands in the channel-octet layout  (round 3; csrc/conv_c8.hip) -----------------------
 * "C8": [n][ceil(C/8)][H][W][8] — the 8 channels of a pixel are one 16-byte entry, which is one entry of the kernel's LDS
 * tile image and one k-octet of an MFMA operand.  Between the convolutions of a dense stack (model/pwc_modules.py:279-286,
 * model/upflow.py:53-60) and of the context network (:396-412) this layout removes the register transposition on the way
 * into LDS and the LDS transposition on the way out, and the tile image is filled by LDS-DMA into the other of two LDS buffers
 * under the matrix phase of the current chunk.  Tensors that other operators write or read plane-wise stay NCHW:
 *   input  = a C8 slice (x8: first octet plane, n8_oct octets; may be absent) followed — in the K order of the packed
 *            weights — by an NCHW part (x2, C2 planes; may be absent);  K = pad32(8*n8_oct) + pad32(C2) = upf_conv_c8_k()
 *   output = C8 octets (y_is_c8; channels that pad the last octet are written as zeros) or NCHW planes.
 * upf_conv_pack_weights_kmap gathers the input channels of w through kmap[K] (-1 = zero) into that K order.
 * stride 1, 16-byte aligned operands; kernel 3x3 with dilation 1 (any layout combination) or 2 / 4 / 8 / 16 (C8 in,
 * C8 out), or 1x1 (NCHW in, C8 out, Cout <= 32); stride 2 for a 3x3 with an NCHW input of > 16 channels and a C8 output of
 * <= 32.  Everything else: upf_conv_forward.
 * Width (round 5): C8 operands take ANY W (a pixel is one 16-byte entry: rows are always aligned); an NCHW input part needs
 * 16-byte aligned rows — W % 8 == 0, or (upf_conv_forward_c8_pitched) a row pitch that is a multiple of 8 with any logical W;
 * an NCHW output may be pitched too (y_row_pitch, 0 = Wo). */
long long upf_conv_packed_bytes_k(int K, int Cout, int kernel_size);
int upf_conv_c8_k(int n8_oct, int C2);
int upf_conv_pack_weights_kmap(const void* w /* [Cout,Cin,k,k] */, void* w_packed, int Cin, int Cout, int kernel_size,
                               const int* kmap /* device, [K] */, int K, int dtype, void* stream);
int upf_conv_forward_c8(const void* x8, long long x8_batch_stride, int n8_oct, const void* x2, long long x2_batch_stride, int C2,
                        const void* w_packed, const float* bias, void* y, long long y_batch_stride, int y_is_c8,
                        int B, int Cout, int H, int W, int kernel_size, int dilation, int stride, float leaky_slope,
                        int dtype, void* stream);
int upf_conv_forward_c8_pitched(const void* x8, long long x8_batch_stride, int n8_oct, const void* x2, long long x2_batch_stride, int x2_row_pitch, int C2,
                                const void* w_packed, const float* bias, void* y, long long y_batch_stride, int y_row_pitch, int y_is_c8,
                                int B, int Cout, int H, int W, int kernel_size, int dilation, int stride, float leaky_slope,
                                int dtype, void* stream);
int upf_conv_c8_set_option(const char* name, int value);   /* "rpw4" (1): Cout <= 32 on large grids: 16-row tiles */
/* Layers with Cout <= 16 (the 2- / 3-channel heads of the dense stacks, 176->8, 160->16; model/pwc_modules.py:262-263,
 * model/upflow.py:36-41) on the 16-output-channel matrix instruction: half the matrix work of a 32-channel block that would be
 * mostly padding.  3x3, dilation 1, stride 1, input = C8 octets only (k-map as in upf_conv_pack_weights_kmap, K % 32 == 0);
 * the same products as upf_conv_forward_c8, summed 32 input channels per instruction instead of 16 (fp32 sums may differ in
 * the last bit). */
long long upf_conv_packed_bytes_k16(int K, int Cout);
int upf_conv_pack_weights_kmap16(const void* w /* [Cout,Cin,3,3] */, void* w_packed, int Cin, int Cout,
                                 const int* kmap /* device, [K] */, int K, int dtype, void* stream);
int upf_conv_forward_c8_narrow(const void* x8, long long x8_batch_stride, int n8_oct, const void* w_packed16, const float* bias,
                               void* y, long long y_batch_stride, int y_is_c8, int B, int Cout, int H, int W, float leaky_slope,
                               int dtype, void* stream);
/* The MERGED NARROW TAIL of a dense stack (round 6).  The layers of a dense stack read nested channel suffixes of one buffer
 * (model/pwc_modules.py:279-286: x = cat([conv_k(x), x]); model/upflow.py:53-60), so the head reads [conv5's output | what
 * conv5 read]:  conv_last = W_last[:, conv5 part] * conv5_out + W_last[:, rest] * rest,  and the second term shares its input
 * with conv5.  A launch with 2 ... 32 output channels costs what the staging of its input costs, so
 *   upf_conv_forward_c8_split: ONE pass over the shared octet input computes a layer of C_main (a multiple of 8) output channels
 *     completely — bias, LeakyReLU, octets into y8 — and, as output channels [C_main, Cout) of the same packed operand (Cout <=
 *     128: they occupy what would be padding of the layer's last 32-channel block(s)), the shared-input part of later layers INCLUDING their bias: raw fp32,
 *     as planes of channel quads partial[n][partial_pitch / 4][y][x][4] (channel c -> float c - C_main; 16 bytes per pixel and quad);
 *   upf_conv_forward_c8_narrow_init: upf_conv_forward_c8_narrow (Cout <= 16) for a later layer on the channels the tail itself
 *     produced, its accumulators starting from floats [partial_offset, +Cout) of that record instead of from a bias.
 * The result is the layer's own fp32 sum in another order; nothing is rounded to 16 bits in between.  3x3, dilation 1, stride 1. */
int upf_conv_forward_c8_split(const void* x8, long long x8_batch_stride, int n8_oct, const void* w_packed /* upf_conv_pack_weights_kmap, Cout rows */,
                              const float* bias /* [Cout] */, void* y8, long long y_batch_stride, int C_main,
                              float* partial, long long partial_batch_stride /* floats */, int partial_pitch /* partial channels per pixel, % 4 == 0 */,
                              int B, int Cout, int H, int W, float leaky_slope, int dtype, void* stream);
int upf_conv_forward_c8_narrow_init(const void* x8, long long x8_batch_stride, int n8_oct, const void* w_packed16,
                                    const float* partial, long long partial_batch_stride, int partial_pitch, int partial_offset /* % 4 == 0 */,
                                    void* y, long long y_batch_stride, int y_is_c8, int B, int Cout, int H, int W, float leaky_slope,
                                    int dtype, void* stream);

/* TWO convolutions as one launch (round 6): a 3x3 layer (Cin <= 16 -> C1 = 16 | 32 channels, LeakyReLU) followed by a 3x3 layer (C1 -> C2 <= 32
 * channels, LeakyReLU) with strides (1, 2) — the two halves of the SGU guidance stem (model/upflow.py:30-33: conv(3, 16), conv(16, 16, stride=2),
 * conv(16, 32), conv(32, 32, stride=2) on both frames at full resolution) — or (2, 1) — the first two stages of the feature pyramid
 * (model/pwc_modules.py:122-142) — whose intermediate (126 / 63 / 31 / 16 MB at 384x1280) then never leaves the chip: a workgroup computes the
 * first layer on the pixels its tile of the second layer reads — rounded to the 16-bit type, zero outside the image: the values the two-launch
 * form reads back — into LDS and multiplies from there.
 * x: NCHW [B,Cin,H,W] with an EVEN row pitch (rows are read as pixel pairs); y: NCHW [B,C2,Ho,Wo] (row pitch given) or, strides (1, 2) only,
 * octets [B,ceil(C2/8),Ho,Wo,8]; Ho = ceil(H/2), Wo = ceil(W/2).  leaky slope 0 = no activation.
 * Packed operands (16-bit, MFMA lane order; lane = co + 32 kg, 8 consecutive k per lane):
 *   wa [S][64][8], S = 3 for Cin <= 4 (k-octet kg of step s = taps 4s + 2kg and 4s + 2kg + 1, 4 channels each), S = 5 for Cin <= 8 (k-octet kg of
 *   step s = tap 2s + kg, channels 0..7), S = 9 for Cin <= 16 (step = tap, kg = channel octet);
 *   wb [9][C1/16][64][8] (tap, k-step of 16 channels, kg = channel octet within the step); rows co >= C / absent taps / channels are zero.
 * The sums are the layers' own fp32 sums in another order than upf_conv_forward's (two taps per instruction in the first layer). */
int upf_conv_pair_forward(const void* x, long long x_batch_stride, int x_row_pitch, int Cin,
                          const void* wa_packed, const float* bias_a /* [C1] */, float slope_a, int C1, int stride_a,
                          const void* wb_packed, const float* bias_b /* [C2] */, float slope_b, int C2, int stride_b,
                          void* y, long long y_batch_stride, int y_row_pitch, int y_is_c8,
                          int B, int H, int W, int dtype, void* stream);

/* ---- the same convolutions under autograd: training on the matrix cores  (model/pwc_modules.py:250-286, :396-412) ----
 * forward      upf_conv_forward on weights packed straight from the fp32 master copy: upf_conv_pack_weights_f32(dgrad=0)
 * data grad    of a stride-1 convolution = the same convolution of grad_y with the flipped, transposed kernel:
 *              upf_conv_pack_weights_f32(dgrad=1) (packs [Cin x Cout], upf_conv_packed_bytes(Cout, Cin, k) bytes), then
 *              upf_conv_forward(grad_pre, ..., Cin := Cout, Cout := Cin, leaky_slope 0, zero bias)
 * activation   upf_leaky_backward: grad_pre = grad_y * (y > 0 ? 1 : slope), y = the forward OUTPUT (n % 8 == 0 elements)
 * weight grad  upf_conv_wgrad: grad_w[co][ci][ky][kx] = sum_{n,y,x} grad_pre[n,co,y,x] * x[n,ci,y+(ky-1)d,x+(kx-1)d], fp32
 *              [Cout,Cin,k,k], an MFMA GEMM with K = pixels, deterministic split-K (workspace:
 *              upf_conv_wgrad_workspace_bytes).  bf16 / fp16, stride 1, W >= 8, 3x3 with d in {1,2,4,8,16} or 1x1
 *              (upf_conv_wgrad_supported); x / grad_pre may be channel slices of wider buffers (batch strides in elements).
 * bias grad    upf_conv_bias_grad: grad_b[co] = sum_{n,y,x} grad_pre[n,co,y,x], fp32, fixed summation order (workspace:
 *              upf_conv_bias_grad_workspace_bytes). */
int upf_conv_pack_weights_f32(const float* w /* [Cout,Cin,k,k] fp32 */, void* w_packed, int Cin, int Cout, int kernel_size,
                              int dtype, int dgrad, void* stream);
/* njobs of the above in one launch (host arrays of the per-layer arguments): a training step re-packs every layer from its
 * fp32 master weights after the optimiser step */
int upf_conv_pack_weights_f32_multi(const float* const* w, void* const* w_packed, const int* Cin, const int* Cout, const int* kernel_size,
                                    const int* dgrad, int njobs, int dtype, void* stream);
/* The data-gradient operands of a DENSE STACK in one launch (round 5).  Layers j = 0..nlayers-1 ([Cout_j, Cin_j, 3, 3] fp32 masters,
 * ordered like the stack's gradient buffer: last layer first) read the buffer channels [first_channel_j, first_channel_j + Cin_j).
 * Slice t = buffer channels [slice_channel_t, + slice_width_t), read by the first nconsumers_t layers: w_packed[t] (upf_conv_packed_bytes(
 * sum Cout_j, slice_width_t, 3) bytes) = upf_conv_pack_weights_f32(dgrad = 1) of the concatenation along Cout of
 * w_j[:, slice_channel_t - first_channel_j : + slice_width_t] — bit-identical, without the concatenation.  nlayers, nslices <= 8. */
int upf_conv_pack_stacked_dgrad(const float* const* w, const int* Cin, const int* Cout, const int* first_channel, int nlayers,
                                void* const* w_packed, const int* slice_channel, const int* slice_width, const int* nconsumers,
                                int nslices, int dtype, void* stream);
int upf_leaky_backward(const void* grad_y, const void* y, void* grad_pre, long long n, float slope, int dtype, void* stream);
int upf_conv_wgrad_supported(int Cin, int Cout, int H, int W, int kernel_size, int dilation, int stride, int dtype);
long long upf_conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int kernel_size, int dilation);
int upf_conv_wgrad(const void* x, long long x_batch_stride, const void* grad_pre, long long g_batch_stride, float* grad_w,
                   void* workspace, int B, int Cin, int Cout, int H, int W, int kernel_size, int dilation, int dtype, void* stream);
/* One weight gradient over SEVERAL uses of the same weights (the decoder is shared by the five pyramid levels,
 * model/upflow.py:535-573): the pixels of every level are one K dimension, so the levels share the K-split launches, the
 * partial blocks and the one ordered reduction (instead of a launch pair per level plus fp32 adds of the results).
 * 1..6 levels of any sizes (each: upf_conv_wgrad_supported); aligned and ragged levels share ONE launch (the staging of a
 * ragged level's border blocks is chosen per tile).  Query the workspace size with the same level array.
 * Environment (A/B runs): UPF_WGRAD_SPLIT="slices,J" overrides the K-split, UPF_WGRAD_ABLATE the experiments of tools/wgrad_ablate.py. */
typedef struct {
  const void* x;        long long x_batch_stride;      /* [B, Cin, H, W] slice, batch stride in elements (0 = dense) */
  const void* grad_pre; long long g_batch_stride;      /* [B, Cout, H, W] slice */
  int B, H, W;
} upf_wgrad_level;
long long upf_conv_wgrad_multi_workspace_bytes(const upf_wgrad_level* levels, int nlevels, int Cin, int Cout, int kernel_size, int dilation);
int upf_conv_wgrad_multi(const upf_wgrad_level* levels /* host array */, int nlevels, float* grad_w, void* workspace, int Cin, int Cout,
                         int kernel_size, int dilation, int dtype, void* stream);
/* ... and, in the same reduction launch, the second stage of the layer's BIAS gradient (upf_conv_bias_grad_finish over
 * 1..8 first-stage buffers; model/pwc_modules.py:250-286 — every decoder layer has a bias): one launch fewer per layer and step.
 * npartials = 0: exactly upf_conv_wgrad_multi. */
int upf_conv_wgrad_multi_bias(const upf_wgrad_level* levels, int nlevels, float* grad_w, void* workspace, int Cin, int Cout,
                              int kernel_size, int dilation, const float* const* bias_partials /* host array */, int npartials, float* grad_bias,
                              int dtype, void* stream);
/* Stride-2 3x3 layers (feature pyramid, SGU guidance; model/pwc_modules.py:95, model/upflow.py:53-55) take the stride-1
 * gradient kernels through their space-to-depth form: xs[(ci,p,q), i, j] = x[ci, 2i+p, 2j+q] (upf_space_to_depth2; inverse = 1
 * for the way back) convolved at stride 1 with a kernel that is w at 9 of its 36 (phase, tap) positions.
 *   weight gradient: upf_conv_wgrad_s2d — levels hold xs ([B, 4*Cin, H/2, W/2]) and grad_pre; grad_w is [Cout, Cin, 3, 3]
 *                    (workspace: upf_conv_wgrad_multi_workspace_bytes with Cin := 4*Cin)
 *   data gradient:   upf_conv_pack_weights_f32(w, ..., dgrad = 2) (upf_conv_packed_bytes(Cout, 4*Cin, 3) bytes), then
 *                    upf_conv_forward(grad_pre, ..., Cin := Cout, Cout := 4*Cin) and upf_space_to_depth2(inverse = 1). */
int upf_space_to_depth2(const void* src, void* dst, int B, int C /* of x */, int H, int W /* of x, even */, int inverse, int dtype, void* stream);
int upf_conv_wgrad_s2d(const upf_wgrad_level* levels, int nlevels, float* grad_w, void* workspace, int Cin, int Cout, int dtype, void* stream);
/* dst = (src + add) * (y > 0 ? 1 : slope) over channel-sliced [B, C, HW] tensors (add, y optional; dst may be src):
 * the gradient entering a layer's pre-activation, with the first stage of the bias gradient (bias_partial: C x 32 fp32,
 * optional) from the same pass; dst = NULL: the bias sums only.  upf_conv_bias_grad_finish sums the first-stage buffers of 1..8 uses in order. */
int upf_act_grad(const void* src, long long src_batch_stride, const void* add, long long add_batch_stride, const void* y,
                 long long y_batch_stride, void* dst, long long dst_batch_stride, float* bias_partial, int B, int C, int HW,
                 float slope, int dtype, void* stream);
/* The 3x3 stride-1 convolution with upf_act_grad's arithmetic in its epilogue (round 5): y = (round16(round16(conv(x)) + add))
 * * (act > 0 ? 1 : mask_slope), bit-identical to upf_conv_forward_pitched (bias as given, no activation) followed by
 * upf_act_grad(src = dst = y, add, y := act, mask_slope).  add and act (either may be NULL, not both): [B,Cout,H,W] channel slices
 * with y's row pitch.
 * The data-gradient convolutions of the dense stacks (reference: autograd through model/pwc_modules.py's estimator / context
 * stacks); their bias sums are then ONE upf_act_grad(dst = NULL) pass over the stack's whole gradient buffer. */
int upf_conv_forward_gated(const void* x, long long x_batch_stride, int x_row_pitch, const void* w_packed, const float* bias,
                           void* y, long long y_batch_stride, int y_row_pitch, const void* add, long long add_batch_stride,
                           const void* act, long long act_batch_stride, float mask_slope,
                           int B, int Cin, int Cout, int H, int W, int dtype, void* stream);
int upf_conv_bias_grad_finish(const float* const* partials /* host array of device pointers */, int npartials, float* grad_bias,
                              int Cout, void* stream);
long long upf_conv_bias_grad_workspace_bytes(int Cout);
int upf_conv_bias_grad(const void* grad_pre, long long g_batch_stride, float* grad_bias, void* workspace, int B, int Cout, int HW,
                       int dtype, void* stream);

/* ---- flow bookkeeping of a pyramid level  (model/upflow.py:566-572) -------------------------------
 * out[n,:] = cast(a + (b + c)) in fp32, b and c optional: `flow_up + res` into the context network's input,
 * `flow_up + (res + fine)` for the next level, or a plain fp32 -> 16-bit copy of a flow into an estimator slot.
 * a : [N,per_item] fp32;  b, c : [N,per_item] of `dtype` (bf16 / fp16) or NULL;  out : fp32 (out_is_f32) or `dtype`,
 * rows out_batch_stride elements apart (0 = per_item). */
int upf_flow_update(const float* a, const void* b, const void* c, void* out, long long out_batch_stride,
                    int out_is_f32, int N, int per_item, int dtype, void* stream);
/* The same sum of a 2-channel flow a : [N,2,HW] (b, c likewise) written as one octet of a C8 buffer: positions 0, 1 = the two
 * components, positions 2..7 = 0.  out8 : the octet, out8_batch_stride in elements. */
int upf_flow_update_c8(const float* a, const void* b, const void* c, void* out8, long long out8_batch_stride,
                       int N, int HW, int dtype, void* stream);

/* ---- occlusion check  (tools.occ_check_model(obj), utils/tools.py:519-588, 641-677) -------------
 * flow_f, flow_b : [B,2,H,W] fp32 -> occ_fw, occ_bw : [B,1,H,W] fp32 in {0,1}. */
int upf_occ_check(const float* flow_f, const float* flow_b, float* occ_fw, float* occ_bw,
                  int B, int H, int W, float alpha1, float alpha2, void* stream);

/* Self-test of the division-free, correctly rounded quotient the sampling kernels use for 2(j+fx)/(W-1) (csrc/sampling.hpp:
 * div_by_const): compares it with the IEEE division over ALL 2^32 bit patterns of the numerator for the divisor
 * max(size-1, 1) and adds the number of differing results to *mismatches (device memory, zeroed by the caller).  ~30 ms. */
int upf_div_selftest(int size, unsigned long long* mismatches, void* stream);

/* ---- soft census distance of the photometric loss  (utils/loss.py:50-91) ----
 * gray1, gray2 : [B,1,H,W] fp32 grey images (0.2989 r + 0.5870 g + 0.1140 b);  dist : [B,1,H,W] fp32,
 *   dist(p) = sum_k d_k/(0.1+d_k),  d_k = (t_k(gray1,p) - t_k(gray2,p))^2,  t_k(I,p) = u/sqrt(0.81+u^2),  u = I(p+k) - I(p),
 * k over the (2*max_distance+1)^2 offsets, images zero padded — the reference's 49-channel identity conv2d + ~10
 * element-wise passes in one launch.  Backward: gradients wrt either grey image (a NULL output is skipped), gather
 * formulation, deterministic. */
int upf_census_forward(const float* gray1, const float* gray2, float* dist, int B, int H, int W, int max_distance, void* stream);
int upf_census_backward(const float* gray1, const float* gray2, const float* grad_dist, float* g_gray1, float* g_gray2,
                        int B, int H, int W, int max_distance, void* stream);

/* ---- loss-side operators of the unsupervised training step  (SURVEY.md §8f rank 3), fp32 ---------------------------
 * boundary-dilated warp, tools.boundary_dilated_warp.warp_im (utils/tools.py:351-499): out[n,c,i,j] = clamp-to-edge
 * bilinear sample of the UN-cropped frame image[n,c] ([B,C,Hi,Wi]) at (j + start[n,0] + flow[n,0,i,j],
 * i + start[n,1] + flow[n,1,i,j]), weights from the clamped corner coordinates.  flow [B,2,h,w], start [B,2] (x, y),
 * out [B,C,h,w].  Backward: gradient wrt the flow (the image is data). */
int upf_boundary_warp_forward(const float* image, const float* flow, const float* start, float* out,
                              int B, int C, int Hi, int Wi, int h, int w, void* stream);
int upf_boundary_warp_backward(const float* image, const float* flow, const float* start, const float* grad_out,
                               float* grad_flow, int B, int C, int Hi, int Wi, int h, int w, void* stream);

/* Deterministic reductions: a forward call writes upf_loss_partials(n_pixels) pairs of floats (one per workgroup) that
 * the caller sums in order; n_pixels = B*H*W. */
int upf_loss_partials(long long n_pixels);

/* 'abs_robust' photometric / distillation term, network_tools.photo_loss_multi_type (model/upflow.py:265-288):
 *   partials[k] = { sum (|x - y| + eps)^q * occ,  sum occ }   over workgroup k's pixels, all C channels;
 * x, y [B,C,HW]; occ [B,HW] or NULL (= 1).  Backward: grad_x = coef[0] * occ * q * (|d|+eps)^(q-1) * sign(d),
 * grad_y = -grad_x (either may be NULL); coef is a DEVICE scalar (upstream gradient / the loss's denominator). */
int upf_robust_loss_forward(const float* x, const float* y, const float* occ, float* partials,
                            int B, int C, int HW, float eps, float q, void* stream);
int upf_robust_loss_backward(const float* x, const float* y, const float* occ, const float* coef,
                             float* grad_x, float* grad_y, int B, int C, int HW, float eps, float q, void* stream);

/* grey = 0.2989 r + 0.5870 g + 0.1140 b of an RGB image [B,3,HW] -> [B,1,HW], the reference's left-to-right evaluation
 * (utils/loss.py:53-55). */
int upf_grey_forward(const float* image, float* grey, int B, int HW, void* stream);

/* Pyramid-distillation term, style 'upup' (model/upflow.py:461-487, network_tools.photo_loss_multi_type 'abs_robust' on
 * upsample_flow(level flow, label)), ONE direction, all levels:
 *   out2[0] = weight * sum_l [ sum (|up(x_l) - y| + eps)^q * occ ] / den,   out2[1] = den = sum occ + 1e-6  (occ NULL: B*2*H*W)
 * x_low[l] [B,2,hs[l],ws[l]] fp32 (1..6 levels, ws >= 2), up = bilinear, align_corners, times the size ratio per component
 * (pwc_modules.py:77-90); y [B,2,H,W] the detached label, occ [B,1,H,W] or NULL.  Forward: one pass over the label + one finishing
 * launch (partials: upf_msd_upup_partials(B,H,W) x 7 floats).  Backward: grad_x_low[l] = d out2[0] / d x_low[l] * grad_out[0],
 * one launch for all levels (deterministic gathers).  x_low / grad_x_low / hs / ws are HOST arrays. */
int upf_msd_upup_partials(int B, int H, int W);
int upf_msd_upup_forward(const float* const* x_low, const int* hs, const int* ws, int nlevels, const float* y, const float* occ,
                         float* partials, float* out2, int B, int H, int W, float weight, float eps, float q, void* stream);
int upf_msd_upup_backward(const float* const* x_low, float* const* grad_x_low, const int* hs, const int* ws, int nlevels, const float* y,
                          const float* occ, const float* grad_out, const float* fwd_out2, int B, int H, int W, float weight, float eps, float q,
                          void* stream);

/* first-order edge-aware smoothness, network_tools.edge_aware_smoothness_order1 (model/upflow.py:197-216):
 *   partials[k] = { sum |pred(i,j)-pred(i+1,j)| * exp(-mean_c |img(i,j)-img(i+1,j)|),  the same along j };
 * loss = sum_x / (B*Cp*(H-1)*W) + sum_y / (B*Cp*H*(W-1)).  img [B,Ci,H,W], pred [B,Cp,H,W].
 * Backward: gradient wrt pred of that loss times the device scalar grad_up[0] (gather, deterministic). */
int upf_smooth_edge1_forward(const float* img, const float* pred, float* partials,
                             int B, int Ci, int Cp, int H, int W, void* stream);
int upf_smooth_edge1_backward(const float* img, const float* pred, const float* grad_up, float* grad_pred,
                              int B, int Ci, int Cp, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UPFLOW_HIP_H */

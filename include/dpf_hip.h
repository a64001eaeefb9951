/* dpf_hip.h -- C ABI of libdpf_hip.so: the MI355X (gfx950) kernels behind the StereoDPNet train/eval path.
 *
 * Boundary conventions (all entry points):
 *   - plain C: device pointers + sizes, no torch / ATen types; float = IEEE fp32; tensors are dense, contiguous
 *     NCHW / NCDHW; `*_host` pointers are HOST memory (small constant tables), everything else is DEVICE memory.
 *   - returns 0 on success, DPF_ERR_INVALID_ARG (-1), DPF_ERR_LAUNCH (-2) or DPF_ERR_UNSUPPORTED (-3).
 *   - no allocation, no host synchronisation; work is enqueued on `stream` (a hipStream_t passed as void*), so a caller
 *     may capture a sequence of calls in a hipGraph.  Scratch comes from the caller (`ws` arguments, sizes below).
 *   - stateless and thread-safe (one stream per call).
 *
 * Each group cites the reference interface (relative to the MinJunKang/DualPixelFace tree) it stands in for.
 */
#ifndef DPF_HIP_H
#define DPF_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DPF_OK 0
#define DPF_ERR_INVALID_ARG (-1)
#define DPF_ERR_LAUNCH (-2)
#define DPF_ERR_UNSUPPORTED (-3)

/* activation codes of dpf_norm_act_* */
#define DPF_ACT_NONE 0
#define DPF_ACT_RELU 1
#define DPF_ACT_PRELU 2
#define DPF_ACT_LEAKY 3
#define DPF_ACT_SIGMOID 4

/* ---- dense convolutions: nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d (cuDNN in the reference) ---------------------
 * call sites: src/module/asm/basics.py:17-36, src/model/stereodpnet/modules.py:26-32,64-69,88-91,208-227,271-296,
 * src/model/stereodpnet/normal_module.py:14-19, src/module/dcn3d/modules/deform_conv.py:310-315 (conv_offset),
 * src/module/asm/asm.py:141-146.  2-D tensors are passed with depth 1 (kd = 1, sd = 1, pd = 0, dd = 1).
 * ws: dpf_conv_workspace_floats(T, reduce_channels, out_channels) floats (repacked weights). */
long long dpf_conv_workspace_floats(int T, int reduce, int outc);
int dpf_conv_forward(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                     int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
                     void* stream);
int dpf_conv_transpose(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                       int K, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                       int dd, int dh, int dw, void* stream);
/* dpf_conv_transpose into an output tensor / weight of Ktot >= K channels, computing only channels [0, K) */
int dpf_conv_transpose_ex(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                          int K, int Ktot, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                          int dd, int dh, int dw, void* stream);
/* dpf_conv_transpose_ex with out += result when accumulate != 0 (data gradients of several convolutions reading one tensor summed in
 * the epilogue); DPF_ERR_UNSUPPORTED with nothing written when the shape does not run on the LDS-DMA kernel */
int dpf_conv_transpose_acc(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                           int K, int Ktot, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                           int dd, int dh, int dw, int accumulate, void* stream);
/* Operand precision of the dense convolution kernels behind dpf_conv_forward* / dpf_conv_transpose* / dpf_conv_wgrad*: 0 = exact fp32
 * (default), 1 = operands rounded to bf16 (round to nearest even) while they are staged, fp32 accumulation and fp32 tensors in HBM --
 * the reference's `precision: 16` (PL autocast, config_/train_faceDP.json) for its nn.Conv2d / nn.Conv3d layers.  Process-wide state;
 * shapes the bf16 kernels do not cover run exact fp32. */
int dpf_set_conv_operand_precision(int bf16);
int dpf_get_conv_operand_precision(void);
/* How operand precision 0 (fp32) multiplies in the stride-1 forward / data-gradient kernels and in the weight-gradient kernel (process-wide;
 * environment DPF_F32_X9 sets the initial value):
 *   2 (default) = each operand as two f16 components of x * 2^s (hi = f16(x 2^s), lo = f16(x 2^s - hi), round to nearest), three partial
 *       products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16, fp32 accumulation.  2^s is a power of two chosen per block and undone
 *       exactly in the epilogue.  A two-component split is exact to 2^-22 only within 2^17 of the scale's maximum, so every kernel GUARDS
 *       the range along the axis its output elements do not sum over (csrc/conv_internal.h): the stride-1 convolutions per POSITION (a
 *       position whose values all lie more than 2^17 below the scale contributes exact zeros and is contracted in an extra pass at its own
 *       scale -- up to three extra passes per channel chunk, the last one takes whatever is left) with one exponent per OUTPUT ROW of the
 *       weights, the weight gradient per CHANNEL (every g row and x channel of a workgroup carries its own exponent), the deformable conv's
 *       gcol products per VOXEL.  With the guards an output element is as accurate, relative to the magnitudes of ITS OWN inputs, as on the
 *       fp32 instruction (tests/test_gpu_ops.py: test_conv_f16_component_path_in_block_dynamic_range and its weight-gradient / deformable
 *       siblings);
 *   1 = the exact round-to-nearest three-way bf16 splits of both operands (x = hi + mid + lo) on the bf16 matrix pipe, the six partial
 *       products that can reach 2^-24 of the product (mid x lo, lo x mid and lo x lo are dropped: <= 2^-23 worst case, rms 2^-26, zero mean);
 *   0 = v_mfma_f32_32x32x2_f32.
 * All three sit at the same distance from an fp64 result (tests/test_gpu_ops.py: test_weight_gradient_f32_matrix_paths_agree,
 * test_conv_f32_matrix_paths_agree, test_conv_f16_component_path_block_scaling). */
int dpf_set_f32_matrix_path(int path);
int dpf_get_f32_matrix_path(void);
/* test aid: 0 switches the position guard of path 2's convolutions off (one scale per channel chunk, no extra passes), 1 (the default
 * and the only setting the product uses) on */
int dpf_debug_set_range_guard(int on);
int dpf_conv_wgrad(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int QD, int QH, int QW,
                   int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_, void* stream);
/* same, with caller scratch `ws` of dpf_conv_wgrad_workspace_floats(T, C, K) floats: eligible shapes (16-byte aligned rows, 3x3 /
 * 3x3x3 kernels, dilation 1) then run without float atomics -- partial tiles are reduced in a fixed order, so the gradient is
 * bitwise reproducible; other shapes use the same kernel as dpf_conv_wgrad.  accumulate = 0: dw is overwritten (any content), 1: dw += */
long long dpf_conv_wgrad_workspace_floats(int T, int C, int K);
int dpf_conv_wgrad_ws(const float* g, const float* x, float* dw, float* ws, long long ws_floats, int N, int C, int ID, int IH, int IW, int K,
                      int QD, int QH, int QW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_,
                      int accumulate, void* stream);

/* dpf_conv_forward + the statistics of the BatchNorm that follows it (convbn / convbn_3d, src/module/asm/basics.py:17-36): per
 * position tile the kernel's epilogue leaves (sum, sum of squares) of every output channel in slab [*parts_host][K][2] doubles
 * (capacity dpf_conv_stats_slab_doubles(...), which includes the scratch rows dpf_bn_finalize_partials folds into behind the tile
 * rows -- pass the same buffer to both); dpf_bn_finalize_partials turns them into mean / invstd / running statistics, so
 * the separate dpf_bn_stats pass over the output is not needed.  Fixed reduction order: bitwise reproducible.
 * DPF_ERR_UNSUPPORTED for shapes the LDS-DMA kernel does not take -- then call dpf_conv_forward + dpf_bn_stats. */
long long dpf_conv_stats_slab_doubles(int N, int K, int OD, int OH, int OW);
int dpf_conv_forward_stats(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                           int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
                           double* slab, long long slab_doubles, int* parts_host, void* stream);
int dpf_bn_finalize_partials(double* slab, int parts, int C, long long count, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean, float* invstd, void* stream);

/* narrow outputs (K <= 4): the 32 -> 1 cost heads (modules.py:286-296) and the 32 -> 3 normal conv (normal_module.py:65);
 * same conventions as dpf_conv_forward / dpf_conv_wgrad, direct (non-MFMA) HBM-bound kernels */
int dpf_conv_smallk_forward(const float* x, const float* w, const float* bias, float* out, int N, int C, int ID, int IH, int IW, int K, int kd,
                            int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, void* stream);
/* data gradient of the same shapes (replaces cuDNN's dgrad behind the 32 -> 1 cost heads, modules.py:331-337, and the 32 -> 3 normal
 * conv): g [N,K,OD,OH,OW], K <= 4, 3x3(x1|x3) window, stride 1, dilation 1, IW % 4 == 0; DPF_ERR_UNSUPPORTED otherwise */
int dpf_conv_smallk_dgrad(const float* g, const float* w, float* dx, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int pd,
                          int ph, int pw, void* stream);
int dpf_conv_smallk_wgrad(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int sd,
                          int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_, void* stream);

/* ---- 2-D convolutions with bf16 operands, fp32 accumulation and storage (v_mfma_f32_32x32x16_bf16): the mixed-precision mode of
 * BASELINE config 5 ("MFMA bf16 2D convs"); the reference's counterpart is PL precision 16 autocast over nn.Conv2d
 * (config_/train_faceDP.json "precision", src/module/asm/basics.py:17-22).  x, w, outputs are fp32 tensors; operands are rounded
 * to bf16 (RNE) inside the kernel.  ws: dpf_conv2d_bf16_workspace_bytes(C, K, kh*kw) bytes.  dgrad: stride-1 convs only. */
long long dpf_conv2d_bf16_workspace_bytes(int C, int K, int T);
int dpf_conv2d_bf16_forward(const float* x, const float* w, const float* bias, float* out, void* ws, int N, int C, int IH, int IW, int K, int kh,
                            int kw, int sh, int sw, int ph, int pw, int dh, int dw, void* stream);
int dpf_conv2d_bf16_dgrad(const float* go, const float* w, float* dx, void* ws, int N, int C, int IH, int IW, int K, int kh, int kw, int ph, int pw,
                          int dh, int dw, void* stream);

/* ---- depthwise 3x3: depthwise_separable_conv.depthwise (src/module/asm/basics.py:39-58) --------------------------- */
int dpf_depthwise_conv2d_forward(const float* x, const float* w, float* y, int N, int C, int H, int W, int k, int pad, void* stream);
int dpf_depthwise_conv2d_backward_data(const float* g, const float* w, float* dx, int N, int C, int H, int W, int k, int pad, void* stream);
int dpf_depthwise_conv2d_backward_weight(const float* g, const float* x, float* dw, int N, int C, int H, int W, int k, int pad, void* stream);

/* ---- BatchNorm2d/3d, InstanceNorm3d, ReLU/PReLU/LeakyReLU/Sigmoid, residual adds ---------------------------------
 * src/module/asm/basics.py:17-58, src/model/stereodpnet/modules.py:37-52,241-260,310-325, src/module/asm/asm.py:138-146.
 * x viewed as [N, C, S].  ws: 2*C floats (stats) / 3*C floats (backward). */
int dpf_bn_stats(const float* x, int N, int C, long long S, float eps, float momentum, float* running_mean, float* running_var,
                 float* mean, float* invstd, float* ws, void* stream);
int dpf_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd, void* stream);
int dpf_norm_act_forward(const float* x, const float* mean, const float* invstd, const float* w, const float* b, int wmod,
                         const float* res, const float* res2, int act, const float* slope, float slope_const, float* y, int N, int C,
                         long long S, void* stream);
int dpf_norm_act_backward(const float* x, const float* dy, const float* mean, const float* invstd, const float* w, const float* b,
                          int wmod, const float* res, int act, const float* slope, float slope_const, int training, float* dx,
                          float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C, long long S, void* stream);
/* the same pair with y / dy given as a channel slice of a wider tensor: the normalised branches of a torch.cat(dim=1) are written
 * straight into the concatenated tensor and their gradients read from its gradient (DPBlock.conv_dilate, modules.py:43-45) */
int dpf_norm_act_forward_slice(const float* x, const float* mean, const float* invstd, const float* w, const float* b, int wmod,
                               const float* res, const float* res2, int act, const float* slope, float slope_const, float* y, int y_channels,
                               int y_c0, int N, int C, long long S, void* stream);
int dpf_norm_act_backward_slice(const float* x, const float* dy, int dy_channels, int dy_c0, const float* mean, const float* invstd,
                                const float* w, const float* b, int wmod, const float* res, int act, const float* slope, float slope_const,
                                int training, float* dx, float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C,
                                long long S, void* stream);
/* SyncBatchNorm (the reference enables torch SyncBatchNorm under DDP: config_manager.py:57, main.py:55).  Statistics: each rank
 * computes dpf_bn_local_moments -> moments[2*C] = {mean, M2}; the host all-gathers them (RCCL) and dpf_bn_merge_moments produces
 * the global mean / invstd and the running-statistics update.  Backward: dpf_norm_act_backward_ex phase 1 (local reductions
 * into ws + parameter gradients), host all-reduce(sum) of ws[3*C], phase 2 (dx / dres) with count = global N*S.
 * phase 0 = everything; phase 3 = everything with a ws the caller guarantees to be zero (no memset: slots of a pre-zeroed arena). */
int dpf_bn_local_moments(const float* x, int N, int C, long long S, float* moments, float* ws, void* stream);
int dpf_bn_merge_moments(const float* moments, const float* counts, int W, int C, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* invstd, void* stream);
int dpf_norm_act_backward_ex(const float* x, const float* dy, const float* mean, const float* invstd, const float* w, const float* b,
                             int wmod, const float* res, int act, const float* slope, float slope_const, int training, float* dx,
                             float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C, long long S, int phase,
                             double count, void* stream);
/* dpf_norm_act_backward_slice with the phases of dpf_norm_act_backward_ex: SyncBatchNorm of branches that wrote channel slices of one
 * concatenated tensor (DPBlock.conv_dilate, modules.py:43-45) -- their exchanges travel in one collective */
int dpf_norm_act_backward_slice_ex(const float* x, const float* dy, int dy_channels, int dy_c0, const float* mean, const float* invstd,
                                   const float* w, const float* b, int wmod, const float* res, int act, const float* slope, float slope_const,
                                   int training, float* dx, float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C,
                                   long long S, int phase, double count, void* stream);
int dpf_channel_sum(const float* g, float* out, int N, int C, long long S, void* stream);

/* ---- resampling: F.interpolate(bilinear, align_corners=True) (modules.py:127-128, normal_module.py:22-29), FPN's
 * nearest top-down add (torchvision.ops.FeaturePyramidNetwork, call site modules.py:83-85,119) ---------------------- */
int dpf_upsample_bilinear2d_forward(const float* x, float* y, long long NC, int h, int w, int H, int W, void* stream);
int dpf_upsample_bilinear2d_backward(const float* g, float* dx, long long NC, int h, int w, int H, int W, void* stream);
/* F.interpolate(size=(H, W), mode='bilinear', align_corners=<flag>): the half-pixel form is used by src/model/nnet/modules.py:110-120 */
int dpf_resize_bilinear2d_forward(const float* x, float* y, long long NC, int h, int w, int H, int W, int align_corners, void* stream);
int dpf_resize_bilinear2d_backward(const float* g, float* dx, long long NC, int h, int w, int H, int W, int align_corners, void* stream);
int dpf_upsample_nearest_add_forward(const float* lat, const float* top, float* y, long long NC, int h, int w, int H, int W, void* stream);
int dpf_upsample_nearest_backward(const float* g, float* dtop, long long NC, int h, int w, int H, int W, void* stream);

/* ---- dual-pixel cost volume: subpixel_shift.forward (src/module/asm/asm.py:87-127), MaskingAttention tail
 * (asm.py:162-171), CostVolume.build_concat_volume (src/model/stereodpnet/modules.py:181-197); PSMNet volume
 * (src/model/psmnet/modules.py:215-262).  Sampler tables: iy/wy [3][2][h], ix/wx [3][2][w] (device). */
int dpf_shift_triple_forward(const float* fea, float* out, const int* iy, const float* wy, const int* ix, const float* wx, int B, int C,
                             int h, int w, void* stream);
int dpf_shift_triple_backward(const float* g, float* dfea, const int* iy, const float* wy, const int* ix, const float* wx, int B, int C,
                              int h, int w, void* stream);
/* deterministic adjoint of dpf_shift_triple_forward in gather form: iy_inv / ix_inv [3][2][2][h|w] are the inverse tap tables (the up to two
 * output coordinates that read a given source coordinate through tap (mode, a), -1 padded), wy / wx as in forward */
int dpf_shift_triple_backward_gather(const float* g, float* dfea, const int* iy_inv, const float* wy, const int* ix_inv, const float* wx,
                                     int B, int C, int h, int w, void* stream);
/* fractional Fourier-phase row shift of `planes` [h, w] planes (asm.py:59-75,112-125; only reached with per-level shifts):
 * dst[p][y][x] = sum_y' mr[(y - y') mod h] src[p][y'][x] + scale (-1)^y sum_x' hm[(x - x') mod w] sum_y' (-1)^y' src[p][y'][x'].
 * mr [h], hm [w]: device tables; tbuf: planes * w floats of scratch; plane strides in floats (dst may be slot 2 of the [B,C,3,h,w]
 * triple).  The adjoint is the same call with index-reversed tables. */
int dpf_phase_shift(const float* src, long long src_plane_stride, float* dst, long long dst_plane_stride, const float* mr, const float* hm,
                    float scale, float* tbuf, long long planes, int h, int w, void* stream);
int dpf_cv_select_forward(const float* x3, const float* s, float* vol, int B, int C, int h, int w, int CV, int L, int choff,
                          unsigned levels, void* stream);
int dpf_cv_select_backward(const float* x3, const float* s, const float* dvol, float* dx3, float* ds, int B, int C, int h, int w, int CV,
                           int L, int choff, unsigned levels, void* stream);
int dpf_psm_volume_forward(const float* ref, const float* tar, float* vol, const int* shifts_host, int B, int C, int h, int w, int L,
                           int groups, void* stream);

int dpf_psm_volume_backward(const float* ref, const float* tar, const float* dvol, float* dref, float* dtar, const int* shifts_host, int B, int C,
                            int h, int w, int L, int groups, void* stream);

/* StereoNet's difference volume (src/model/stereonet/mainmodel.py:97-112): vol [B, C, L, h, w] = ref - target shifted by shifts_host[l] rows on
 * the rows the reference writes, 0 elsewhere; and its adjoint */
int dpf_diff_volume_forward(const float* ref, const float* tar, float* vol, const int* shifts_host, int B, int C, int h, int w, int L,
                            void* stream);
int dpf_diff_volume_backward(const float* dvol, float* dref, float* dtar, const int* shifts_host, int B, int C, int h, int w, int L,
                             void* stream);

/* ---- nn.AvgPool2d(k, stride k) of PSMNet's SPP branches (src/model/psmnet/modules.py:84-102) ---------------------- */
int dpf_avg_pool2d_forward(const float* x, float* y, long long NC, int H, int W, int k, void* stream);
int dpf_avg_pool2d_backward(const float* g, float* dx, long long NC, int H, int W, int k, void* stream);

/* ---- disparity head: trilinear x4 + softmax + soft-argmin (modules.py:327-334,341-362) ---------------------------- */
int dpf_softargmin_forward(const float* logits, float* pred, float* prob, const float* disp_host, int B, int D, int h, int w, int L,
                           int H, int W, void* stream);
int dpf_softargmin_backward(const float* logits, const float* gpred, float* dlogits, const float* disp_host, int B, int D, int h, int w,
                            int L, int H, int W, void* stream);

/* the same head with the x4 trilinear upsampling in either convention (align_corners = 0: src/model/nnet/mainmodel.py:150-153) */
int dpf_softargmin_forward_ex(const float* logits, float* pred, float* prob, const float* disp_host, int B, int D, int h, int w, int L,
                              int H, int W, int align_corners, void* stream);
/* forward with the probability volume written at a batch stride: head i of n fills slice [:, i] of a [B, n, L, H, W] tensor */
int dpf_softargmin_forward_strided(const float* logits, float* pred, float* prob, long long prob_batch_stride, const float* disp_host, int B,
                                   int D, int h, int w, int L, int H, int W, int align_corners, void* stream);
int dpf_softargmin_backward_ex(const float* logits, const float* gpred, float* dlogits, const float* disp_host, int B, int D, int h, int w,
                               int L, int H, int W, int align_corners, void* stream);

/* ---- deterministic mode (SURVEY section 5 "race detection"; the reference scatters with float atomics, deform_im2col_cuda.cuh:313-331, and
 * is not reproducible either).  0 (default; environment DPF_DETERMINISTIC=1 changes the default): partial results of overlapping tiles /
 * workgroups are merged with float atomics where that is fastest.  1: every such merge is order-independent -- one committing workgroup per
 * output address (BatchNorm statistics and backward sums, bias gradients, depthwise / first-generation weight gradients, the loss sums),
 * phased launches of tiles with disjoint footprints + integer LDS cells (soft-argmin head backward), integer accumulation (deformable conv
 * grad_input through a shadow tensor, grad_weight scratch) -- so two runs of the same step produce the same bits.  Slower.  Process-wide. */
int dpf_set_deterministic(int on);
int dpf_get_deterministic(void);
/* measurement aid: one lane samples {shader cycle counter, 100 MHz counter} at the start and after `spin_us` microseconds into out4 (device,
 * 4 x 64 bit).  Launched on a side stream while the kernels of interest run on another, it reports the shader clock the chip holds under
 * that load: MHz = (out4[2] - out4[0]) / (out4[3] - out4[1]) * 100 (bench.py "shader_clock_mhz_under_conv_load"). */
int dpf_debug_clock_probe(unsigned long long* out4, int spin_us, void* stream);

/* ---- deformable conv3d: the reference's pybind module `DCN` (src/module/dcn3d/src/vision.cpp:4-7,
 * src/module/dcn3d/src/deform_conv.h:10-29,49-69; deform_conv_cuda.cu:18-285) -- same argument order and meaning ----- */
long long dpf_deform_conv3d_workspace_floats(int C, int K, int T);
/* workspace of dpf_deform_conv3d_backward* for this problem: the above, plus (deterministic mode only) the integer shadow of grad_input */
long long dpf_deform_conv3d_backward_workspace_floats(int B, int C, int D, int H, int W, int K, int T);
int dpf_deform_conv3d_forward(const float* input, const float* weight, const float* bias, const float* offset, float* output, float* ws,
                              int B, int C, int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph,
                              int pw, int dd, int dh, int dw, int group, int deformable_group, int im2col_step, void* stream);
int dpf_deform_conv3d_backward(const float* input, const float* weight, const float* bias, const float* offset, const float* grad_output,
                               float* grad_input, float* grad_offset, float* grad_weight, float* grad_bias, float* ws, int B, int C,
                               int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
                               int dh, int dw, int group, int deformable_group, int im2col_step, void* stream);

/* extension: grad_input only for input channels [0, grad_input_channels) (the rest of the zero-initialised tensor stays 0) */
int dpf_deform_conv3d_backward_ex(const float* input, const float* weight, const float* bias, const float* offset, const float* grad_output,
                                  float* grad_input, float* grad_offset, float* grad_weight, float* grad_bias, float* ws, int B, int C,
                                  int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
                                  int dh, int dw, int group, int deformable_group, int im2col_step, int grad_input_channels, void* stream);

/* ---- Adaptive Normal Module glue (src/model/stereodpnet/normal_module.py:80-138,154-167,185-190) ------------------ */
int dpf_anm_select(const float* disp_full, int* idx, float* sdisp, const float* costrange_host, int B, int H, int W, int h, int w, int L,
                   int K, void* stream);
int dpf_anm_volume_forward(const float* cost, const int* idx, const float* sdisp, const float* Kmat, const float* abvalue, float* vol,
                           unsigned* mm_ws, int B, int C, int L, int K, int h, int w, void* stream);
int dpf_anm_volume_backward(const float* dvol, const int* idx, float* dcost, int B, int C, int L, int K, int h, int w, void* stream);
/* camera-space coordinate volume alone (NNet's plain normal module, src/model/nnet/normal_module_.py:50-87; same arithmetic as the XYZ
 * channels of dpf_anm_volume_forward): channels [choff, choff + 3) of vol [B, CV, K, h, w]; sdisp [B, K, h, w]; mm_ws: 2 * B ints */
int dpf_xyz_volume(const float* sdisp, const float* Kmat, const float* abvalue, float* vol, int* mm_ws, int B, int choff, int CV, int K, int h,
                   int w, void* stream);
/* F.normalize(x, dim=1) for x [N, C, S] (normal_module_.py:114) and its gradient */
int dpf_l2_normalize_forward(const float* x, float* y, int N, int C, long long S, float eps, void* stream);
int dpf_l2_normalize_backward(const float* x, const float* g, float* dx, int N, int C, long long S, float eps, void* stream);
int dpf_sigmoid_mean_forward(const float* u, float* out, int B, int Dn, long long CS, void* stream);
int dpf_sigmoid_mean_backward(const float* u, const float* g, float* du, int B, int Dn, long long CS, void* stream);

/* ---- tensor plumbing kept off eager PyTorch: channel-range copies (torch.cat / stack / slices: modules.py:42,129,
 * mainmodel.py:98-100), [B,A,Bd,S] -> [B,Bd,A,S] (normal_module.py:185,187), channel max (mainmodel.py:104), closed-form
 * replay of the shared attention BatchNorm's running statistics (asm.py:141-146 called 2*level times) ----------------- */
int dpf_copy_channels(const float* src, float* dst, int N, int Cs, int cs0, int Cd, int cd0, int ncopy, long long S, int accumulate,
                      void* stream);
int dpf_swap_axes(const float* src, float* dst, int B, int A, int Bd, long long S, void* stream);
int dpf_channel_max(const float* x, float* y, int N, int C, long long S, void* stream);
int dpf_bn_replay(float* running, const float* a_f, const float* a_b, int C, float decay, float cf, float cb, void* stream);

/* ---- losses (src/loss/loss_selector.py:29-42, src/loss/depth/smoothL1.py:15-49, src/loss/normal/cosine.py:15-53)
 * and Adam (src/model/model_selector.py:31-34) --------------------------------------------------------------------- */
int dpf_loss_forward(const float* pred_depth, const float* pred_normal, const float* disp, const float* normal, const float* mask,
                     float* acc_ws, float* out, int B, int n, int H, int W, const float* head_weights_host, float lambda_depth,
                     float lambda_normal, void* stream);
int dpf_loss_backward(const float* pred_depth, const float* pred_normal, const float* disp, const float* normal, const float* mask,
                      const float* acc_ws, const float* gout, float* d_pred_depth, float* d_pred_normal, int B, int n, int H, int W,
                      const float* head_weights_host, float lambda_depth, float lambda_normal, void* stream);
int dpf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, int step, double lr, double beta1,
                  double beta2, double eps, float gscale, void* stream);
/* the same update with the two step-dependent scalars { (float)(lr / (1 - beta1^step)), (float)(1 / sqrt(1 - beta2^step)) } read from device
 * memory (hyper[2]): a captured HIP graph of the whole train step replays this launch unchanged while the host refreshes the two floats */
int dpf_adam_step_hyper(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, const float* hyper, double beta1,
                        double beta2, double eps, float gscale, void* stream);

/* ---- FaceDP sample preprocessing: the per-sample host work of the reference's DataLoader workers,
 * dataloader/FaceDP/path_reader.py:150-168 (read_depth), :196-232 (read_disparity), dataloader/preprocess/preprocess.py:46-88
 * (basic_transform.apply), dataloader/preprocess/augmentation.py:62-84 (to_tensor step), :165-178 (crop), :236-262 (Lighting),
 * :265-297 (Normalizer).  The host keeps file IO, JPEG decoding and the random draws; the decoded full-resolution arrays are
 * uploaded once and every crop / raw view is produced from them on the device.
 *
 * depth: [H, W] metric depth, fp32 (depth_f64 = 0) or fp64 (depth_f64 = 1), as stored in the .npy; mask: optional u8 [H, W]
 * (file mask > 0), NULL -> depth > 0.  a, b: disparity = a / depth + b (abvalue_list[camidx] = [a, b]).
 * stats: dpf_dp_stats_doubles() doubles of device scratch; after dpf_dp_depth_stats stats[0] = max valid depth, stats[1] = max
 * valid disparity (fp64, NaN-propagating like np.max), stats[3] = number of valid pixels; dpf_dp_targets adds to stats[2] the
 * number of written pixels whose disparity or inverse depth is not finite (the reference raises on those). */
long long dpf_dp_stats_doubles();
int dpf_dp_depth_stats(const void* depth, int depth_f64, const unsigned char* mask, long long n, double a, double b, double* stats,
                       void* stream);
/* window [y0, y0 + ch) x [x0, x0 + cw) -> depth_out / mask_out / disp_out fp32 [ch, cw], idepth_out [ch, cw] in the depth's type.
 * disp = fp32(a / fp64(depth) + b), 50 * stats[1] on invalid / NaN / Inf pixels; idepth = max_depth / depth, 0 outside the mask;
 * depth 0 outside the mask; any output may be NULL */
int dpf_dp_targets(const void* depth, int depth_f64, const unsigned char* mask, double* stats, double a, double b, int H, int W, int y0,
                   int x0, int ch, int cw, float* depth_out, float* mask_out, float* disp_out, void* idepth_out, void* stream);
/* img u8 [H, W, C] (C = 1 or 3, 4-byte aligned) window -> out fp32 [C, ch, cw] = ((lut_c[v] / 255 + shift[c]) - mean[c]) / std[c], each
 * step rounded to fp32 (to_tensor, Lighting, Normalizer).  lut: optional device u8 [C, 256] photometric table (brightness / gamma /
 * contrast on 8-bit values), NULL = identity; shift_host NULL = 0; mean 0 / std 1 gives the reference's raw_transform */
int dpf_dp_image(const unsigned char* img, const unsigned char* lut, float* out, int H, int W, int C, int y0, int x0, int ch, int cw,
                 const float* shift_host, const float* mean_host, const float* std_host, void* stream);
/* fp32 [H, W, C] window -> [C, ch, cw] (normal / albedo maps through to_tensor) */
int dpf_dp_hwc_to_chw(const float* src, float* dst, int H, int W, int C, int y0, int x0, int ch, int cw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPF_HIP_H */

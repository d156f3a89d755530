/* libonepose_hip.so -- C ABI of the MI355X (gfx950) implementation of the OnePose++ 2D-3D matching hot path.
 *
 * The reference (mizeller/OnePose_ST) has no FFI/plugin boundary: the path sits behind a Python nn.Module
 * (src/models/OnePosePlus/OnePosePlusModel.py:24-203).  This header is the boundary a maintainer binds with
 * ctypes; each entry point replaces the reference sub-module named in its comment.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host"; plain pointers and sizes only
 *   - float = IEEE f32, ids = int64 (torch.long), all tensors dense row-major in the stated shape
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, never synchronise,
 *     never allocate: callers pass workspaces sized by the *_workspace_floats() helpers
 *   - return 0 on success; -1 invalid argument; otherwise the hipError_t.  ophip_last_error() (host
 *     string, thread local) describes the last failure.
 *   - packed weight blocks are produced by onepose_st_amd/packing.py (layout documented there and in
 *     csrc/tile.h): W[out][in] in MFMA fragment order [out/32][in/8][64 lanes][4].
 */
#ifndef ONEPOSE_HIP_H
#define ONEPOSE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Version of THIS header's structs and signatures; ophip_abi_version() returns the value the library was built with and a caller
 * compares the two before its first call.  History: 1 = rounds 1-2; 2 = round 3 (lazy_conf inside ophip_frame_desc, the extended
 * ophip_frame_layout_t, ophip_frame_wait takes the TICKET ophip_frame_enqueue returned (generation * 16 + slot, never below 16),
 * ophip_encoder_layer_x3 / ophip_fine_refine_x3 removed); 3 = round 4 (ophip_encoder_kv_first_x3w8 added, kv_from_prev = 2 accepted by the
 * x3w8 layer entry points: additive, but a binding that names the new symbol needs a library that has it); 4 = round 5 (the object cache:
 * ophip_object_cache, ophip_encoder_object_x3w8, ophip_encoder_x3w8_kv_block_bytes, ophip_frame_enqueue_object; ophip_frame_layout's
 * external_x3d takes 2 = "first layer cached as well"; ophip_frame_enqueue{,_padded} with an x3d_external issue the first layer's K / V
 * half on s_main: additive otherwise). */
#define OPHIP_ABI_VERSION 4
int ophip_abi_version(void);
/* host string: 16 hex digits of the sha256 over the sources this library was built from (the profiles/ pmc summaries record it;
 * bench.py quotes committed counter values only when they were taken on the running build) */
const char* ophip_build_stamp(void);
const char* ophip_last_error(void);
/* Tracing hook (the reference's profiler.record_function scopes, coarse_matching.py:122,167): on = 1 wraps every kernel
 * launch and every stage of ophip_frame_enqueue in a roctx range (rocprofv3 --marker-trace); OPHIP_ROCTX=1 in the environment enables it
 * without a call.  libroctx64 is loaded at run time; an error is returned when it is missing.  ophip_roctx_ranges(): ranges opened so far. */
int ophip_roctx_enable(int on);
long long ophip_roctx_ranges(void);
/* host query: CU count, LDS bytes per block, gcn arch name of the current device */
int ophip_device_info(int* cu_count, int* lds_per_block, char* arch, int arch_len);

/* Measurement hook (bench.py): select one kernel by its launch name ("attn_apply", "kv_reduce", "kv_sum", "frag_planes", "sim_stats" (the
 * similarity tiles: sim_frag / sim_stats kernels), "stat_combine", "conf", "select" (select_decide), "select_place", "fine_refine",
 * "pe_add_transpose", "transpose_cl", "kpt_stats", "kpt_encode", "stem", "conv", "crop_resize", "fine2_gather", "fine2_attention",
 * "fine2_match", "rows_linear", "rows_layernorm"; "" = off).
 * While selected, every launch of that kernel is bracketed by a hipEvent pair on its launch stream (at most 8192
 * launches between reads).  ophip_timing_read() synchronises those events, returns the launch count and the summed
 * device time, and clears the log. */
int ophip_timing_select(const char* kernel_name);
int ophip_timing_read(int* launches, double* total_ms);
/* bracket only every n-th launch of the selected kernel (default 1): an event pair costs ~2-3 us of stream time */
int ophip_timing_every(int n);

/* Diagnostics only: while a device buffer (32 x uint64 per workgroup of the largest instrumented launch) is set, the
 * fine_refine / attn_apply bf16 kernels record s_memtime at their phase boundaries into it.  NULL switches it off. */
int ophip_debug_stamps(void* device_buffer);

/* a1 -- PositionEncodingSine.forward + rearrange 'n c h w -> n (h w) c'
 * (utils/position_encoding.py:37-42, OnePosePlusModel.py:135-140).
 * out[b][m][c] = feat[b][c][m] + pe_nlc[m][c];  pe_nlc may be NULL (positional encoding disabled). */
int ophip_pe_add_transpose(const float* feat_nchw, const float* pe_nlc, float* out_nlc, int B, int C, int M, void* stream);

/* [B][C][L] -> [B][L][C]  (the einsum 'bdn->bnd' of transformer.py:145 when keypoint encoding is disabled) */
int ophip_transpose_cl(const float* in_bcl, float* out_blc, int B, int C, int L, void* stream);

/* a2 + a3 -- normalize_3d_keypoints + KeypointEncoding_linear.forward, emitted token-major
 * (utils/normalize.py:17-28, utils/position_encoding.py:54-79, transformer.py:145).
 * keypoints3d [B][N][3] (batch stride kpts_bstride floats; 0 = shared object block),
 * desc_bcn [B][256][N] (batch stride desc_bstride), stats: scratch of 4*B+4 floats,
 * out_bnc [B][N][256].  wpack: keypoint-encoder block of packing.py. */
int ophip_kpt_encode(const float* keypoints3d, long long kpts_bstride, const float* desc_bcn, long long desc_bstride,
                     const float* wpack, float* stats, float* out_bnc, int B, int N, void* stream);

/* a4-a6 -- one LoFTREncoderLayer applied to both streams of the coarse encoder (d_model 256, 8 heads)
 * (loftr_module/transformer.py:65-94 and :146-159, loftr_module/linear_attention.py:29-61).
 * x3d [B][L3d][256], x2d [B][L2d][256] -> y3d, y2d (must not alias the inputs: both streams of a layer read
 * the pre-update tensors).  is_cross: 0 = "self", 1 = "cross".  workspace: ophip_encoder_workspace_floats(). */
size_t ophip_encoder_workspace_floats(int B, int L3d, int L2d);
int ophip_encoder_layer(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                        const float* wpack, int is_cross, float* workspace, void* stream);

/* Same layer on the bf16 matrix pipe with PLAIN bf16 operands (v_mfma_f32_32x32x16_bf16, f32 accumulate; the "bf16"
 * arithmetic mode).  nsplit must be 1 (the split-bf16 layer is ophip_encoder_layer_x3w8).  wpack: packing.pack_coarse_layer_bf16
 * (ophip_encoder_bf16_wpack_bytes() bytes); workspace: ophip_encoder_bf16_workspace_bytes() bytes, 256-byte aligned.
 * Layer chaining: when wpack_next (the NEXT layer's block) is given, attn_apply also produces that layer's K/V partial
 * slabs from the output tile while it is still on chip; the next call then passes kv_from_prev = 1 (skips its own
 * kv_reduce launch) and slot ^ 1 (the two slab sets of the workspace ping-pong).  A stand-alone layer: wpack_next = NULL,
 * kv_from_prev = 0, slot = 0. */
size_t ophip_encoder_bf16_workspace_bytes(int B, int L3d, int L2d);
size_t ophip_encoder_bf16_wpack_bytes(void);
int ophip_encoder_layer_bf16(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                             const void* wpack, const void* wpack_next, int nsplit, int is_cross, int kv_from_prev, int slot,
                             void* workspace, void* stream);

/* The default layer: split-bf16 (x = hi + lo, three MFMAs per product, f32 accumulate), v_mfma_f32_16x16x32_bf16 on 48-token
 * workgroups of eight waves (one workgroup per CU, two waves per SIMD; csrc/encoder_x3w8.hip), the layer's weights pre-ordered
 * into one linear stream per wave (packing.pack_coarse_layer_x3w8, ophip_encoder_x3w8_wpack_bytes() bytes, 16-byte aligned),
 * epilogues overlapped with the next GEMM inside a wave.  Arguments and layer chaining (wpack_next / kv_from_prev / slot) as
 * ophip_encoder_layer_bf16; workspace: ophip_encoder_x3w8_workspace_bytes() bytes.
 * Replaces transformer.py:65-94 + linear_attention.py:29-61. */
size_t ophip_encoder_x3w8_workspace_bytes(int B, int L3d, int L2d);
size_t ophip_encoder_x3w8_wpack_bytes(void);
/* ..._frag: the same layer; the output rows are also written as the similarity kernel's operand fragments (see
 * ophip_coarse_frag_planes) -- for the LAST layer of the encoder. */
int ophip_encoder_layer_x3w8_frag(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                  const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                                  void* workspace, void* frag3d, void* frag2d, void* stream);
int ophip_encoder_layer_x3w8(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                             const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                             void* workspace, void* stream);
/* One layer on a SUBSET of the two streams' rows: streams bit 0 = the first stream (x3d -> y3d), bit 1 = the second (x2d -> y2d); the
 * layer projects its own K / V (of the stream(s) the running rows attend to) and has no fused tail.  For LoFTR's sequential cross layers
 * (feat0 = layer(feat0, feat1); feat1 = layer(feat1, feat0_new): one launch per image instead of two two-stream launches that each
 * discard half their rows).  The rows that run equal the same rows of ophip_encoder_layer_x3w8 bit for bit; the pointers of a stream
 * that neither runs nor is attended to may be NULL.  Replaces transformer.py:65-94 on one stream. */
int ophip_encoder_layer_x3w8_streams(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                     const void* wpack, int is_cross, int streams, void* workspace, void* stream);
/* The K / V half of a layer that projects its own K, V (the FIRST layer of a frame; transformer.py:65-94 k_proj / v_proj +
 * linear_attention.py:49-57): K, V projections of both streams -> phi(K)^T V / Ksum slabs -> their fixed-order sum into the workspace.
 * It reads the layer's input rows only, so a frame pipeline issues it as soon as those exist -- beside whatever the previous frame
 * still runs -- and then calls ophip_encoder_layer_x3w8{,_frag,_masked} with kv_from_prev = 2 ("the summed block is there": same wpack,
 * slot, workspace).  mask2d: NULL, or the layer's query mask.  Bit-identical to the one-call layer (same kernels in the same order). */
int ophip_encoder_kv_first_x3w8(const float* x3d, const float* x2d, int B, int L3d, int L2d, const void* wpack, int slot,
                                void* workspace, const unsigned char* mask2d, void* stream);

/* a7 + a8 -- CoarseMatching.forward + get_coarse_match, inference branch
 * (utils/coarse_matching.py:76-123, :125-242, mask_border :10-20).
 * feat3d [B][N][256], feat2d [B][M][256] (encoder outputs), M = hc * wc.
 * conf [B][N][M] receives the dual-softmax confidence matrix (data["conf_matrix"]).
 * Outputs (capacity B*N entries each, ascending (b, i)): b_ids/i_ids/j_ids int64, mconf, mkpts3d [.][3]
 * (= keypoints3d[b, i]), mkpts_c [.][2] (= (j % wc, j / wc) * scale); *count = K.
 * m_bids (int64, = b_ids) and gt_mask (one byte per match, mconf == 0) are the two remaining keys of the reference's
 * coarse_matches dict (coarse_matching.py:228-240); either may be NULL.
 * temperature is passed as double so that (float)(temperature + 1e-4) matches the reference's scalar.
 * nsplit selects the arithmetic of the similarity GEMM: 0 exact f32 MFMA, 1 bf16, 3 split-bf16.
 *
 * LAZY conf_matrix ("may be produced lazily"; the inference callers read only the match lists, inference.py:179-180):
 * conf == NULL in a bf16 mode.  Nothing N x M is stored: a first pass over the similarity tiles leaves the softmax statistics, a
 * second one recomputes every tile, forms the confidences with conf_matrix's own expression (bit-identical values) and keeps per
 * (row, column tile) the best candidate above the threshold and the column maxima -- what the selection consumes.  Indices and
 * mconf are bit-identical to the eager form.  `count` must then point at TWO ints: count[1] is set to 1 when a row's maximum is
 * tied exactly between columns and the first of them fails the mutual test (resolving that needs the stored row): the caller
 * re-runs the frame with a conf buffer.  count[1] = 0 otherwise. */
size_t ophip_coarse_workspace_floats(int B, int N, int M);
/* bf16 modes: the similarity kernel reads its operands as (hi, lo) bf16 MFMA fragments that a first kernel derives from
 * feat3d / feat2d.  A producer that writes them itself (ophip_encoder_layer_x3w8_frag) gets their place inside the workspace
 * from ophip_coarse_frag_planes() and passes nsplit | OPHIP_COARSE_PLANES_READY: that kernel is then skipped. */
#define OPHIP_COARSE_PLANES_READY 0x100
int ophip_coarse_frag_planes(float* workspace, int B, int N, int M, void** planes3d, void** planes2d);
int ophip_coarse_match(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                       int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                       float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                       float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                       int* count, int nsplit, void* stream);
/* The same call in two halves, same arguments: _conf runs everything up to and including the conf_matrix store and the
 * per-row / per-column best candidates (wide kernels); _select runs the single-workgroup mutual-NN selection that writes the
 * match lists and *count.  _select only feeds the fine stage and the read-back, so a caller may issue it on the fine
 * stage's stream (after an event on _conf) and let the next frame's input kernels run beside it. */
int ophip_coarse_match_conf(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                            int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                            float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                            float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                            int* count, int nsplit, void* stream);
int ophip_coarse_match_select(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                              int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                              float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                              float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                              int* count, int nsplit, void* stream);

/* a9 + a10 + a11 -- FinePreprocess + fine LocalFeatureTransformer (d_model 128) + FineMatching
 * (loftr_module/fine_preprocess.py:32-55, loftr_module/transformer.py:133-171, utils/fine_matching.py:28-110).
 * feat_f: fine feature map addressed by strides in floats (NCHW or channels-last), hf x wf.
 * desc3d_f [B][128][N] (strides ds_b, ds_c).  Matches come from ophip_coarse_match (device count, no host sync);
 * the grid covers max_matches, surplus workgroups exit.  expec_f [.][3] = (x, y, std), mkpts_f [.][2].
 * dbg_win [.][25][128] / dbg_f3 [.][128] (both or neither; NULL in production) receive the fine-encoder outputs. */
int ophip_fine_refine(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                      const float* desc3d_f, long long ds_b, long long ds_c,
                      const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                      const float* mkpts_c, const float* wpack, int nlayers, unsigned cross_bits, int encoder_enable,
                      int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                      float* dbg_win, float* dbg_f3, void* stream);

/* ophip_fine_refine on the bf16 matrix pipe (nsplit = 1 plain bf16, 3 split-bf16); one match per 4-wave workgroup.
 * wpack: packing.pack_fine_layers_bf16 (ophip_fine_bf16_wpack_bytes(nlayers) bytes).  Other arguments as above. */
size_t ophip_fine_bf16_wpack_bytes(int nlayers);
int ophip_fine_refine_bf16(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                           const float* desc3d_f, long long ds_b, long long ds_c,
                           const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                           const float* mkpts_c, const void* wpack, int nlayers, unsigned cross_bits, int encoder_enable, int nsplit,
                           int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                           float* dbg_win, float* dbg_f3, void* stream);

/* Padded / resized query images (datasets with img_pad / img_resize; the demo config has neither).  The reference's forward takes
 * two optional inputs (OnePosePlusModel.py:104,156-158):
 *   data["query_image_mask"] [B][hc][wc] bool, 1 = real cell, 0 = padding  -> query_mask [B][M] bytes (flatten(-2));
 *   data["query_image_scale"] [B][2] f32, (h, w) factors original / resized -> query_scale.
 * query_mask: in the coarse encoder the 2D stream's padded rows are masked as queries AND as sources (transformer.py:148-159 passes it
 * as x_mask / source_mask, linear_attention.py:49-53 zeroes phi(Q) resp. phi(K), V of those rows; v_length stays the padded length);
 * in coarse matching -1e9 is added to the padded cells' columns of the similarity (coarse_matching.py:108-114: their confidences
 * are exactly 0).  query_scale: mkpts_query_c = (x, y) * scale * query_scale[b][[1, 0]] (coarse_matching.py:224) and the fine offset
 * is multiplied by the same factors (fine_matching.py:104).  The *_masked / *_scaled entry points below are the plain ones with
 * these pointers added; the mask / scale must not be NULL in the encoder / fine variants (call the plain entry point instead),
 * either may be NULL in ophip_coarse_match_masked (parts: 3 = whole stage, 1 = the _conf half, 2 = the _select half). */
int ophip_encoder_layer_masked(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                               const float* wpack, int is_cross, float* workspace, const unsigned char* query_mask, void* stream);
int ophip_encoder_layer_bf16_masked(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                    const void* wpack, const void* wpack_next, int nsplit, int is_cross, int kv_from_prev, int slot,
                                    void* workspace, const unsigned char* query_mask, void* stream);
int ophip_encoder_layer_x3w8_masked(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                    const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                                    void* workspace, const unsigned char* query_mask, void* stream);
int ophip_coarse_match_masked(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                              int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                              float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                              float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                              int* count, int nsplit, int parts, const unsigned char* query_mask, const float* query_scale, void* stream);
int ophip_fine_refine_scaled(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                             const float* desc3d_f, long long ds_b, long long ds_c,
                             const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                             const float* mkpts_c, const float* wpack, int nlayers, unsigned cross_bits, int encoder_enable,
                             int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                             float* dbg_win, float* dbg_f3, const float* query_scale, void* stream);
int ophip_fine_refine_bf16_scaled(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                                  const float* desc3d_f, long long ds_b, long long ds_c,
                                  const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                                  const float* mkpts_c, const void* wpack, int nlayers, unsigned cross_bits, int encoder_enable, int nsplit,
                                  int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                                  float* dbg_win, float* dbg_f3, const float* query_scale, void* stream);

/* One call per frame -- OnePosePlus_model.forward after the backbone (OnePosePlusModel.py:115-203), default split-bf16 path:
 * rows a1-a11 on three streams (input kernels | encoder + coarse matching | selection + fine stage) and the read-back of the
 * result block, issued from C (csrc/frame.hip) instead of ~27 separate calls from the host language.
 * ophip_frame_desc: sizes, scalars and packed-weight pointers of the model (fixed per model and input shape).
 * ophip_frame_layout(): byte offsets of every intermediate and output inside ONE device block of `total` bytes
 *   (transpose_fine: the fine map arrives NCHW and needs the channels-last copy; external_x3d: 1 = the keypoint encoding is
 *   supplied by the caller -- a cached object block -- and stays read-only; 2 = so are the first encoder layer's 3D rows, see
 *   ophip_object_cache below).
 * ophip_frame_enqueue(): enqueues the frame and returns; nothing is allocated or synchronised.  s_prep may be NULL (input
 *   kernels then run on s_main, behind everything queued there); host_dst receives the first host_bytes of the result block
 *   (>= 16: the match count; result_bytes: count | b_ids | mkpts3d | mkpts2d) by an asynchronous copy on s_copy.
 *   The block, the inputs and host_dst must stay valid until ophip_frame_wait(*slot) returned.  *slot receives a TICKET
 *   (generation * 16 + ring index): the 16 event sets are reused round-robin; before one is reused enqueue waits for the frame
 *   that held it, and ophip_frame_wait() of a ticket whose set has been handed out again returns at once (that frame is done).
 * Result block at offset `result`: int32 count @0, int64 b_ids[cap] @16, float mkpts3d[cap][3], float mkpts_query_f[cap][2],
 *   cap = B * N; the other outputs (conf_matrix, i_ids, j_ids, m_bids, gt_mask, mconf, mkpts_query_c, expec_f) at their offsets.
 *   Pipelining: when s_fine is a stream of its own, the fine stage (a9-a11) and the read-back of frame t are kept back and
 *   launched by the NEXT ophip_frame_enqueue() on the same s_main, behind that frame's similarity tiles -- they then run beside its
 *   HBM-bound confidence pass instead of in front of it (OPHIP_FRAME_DEFER_FINE=0: every frame in order).  A
 *   frame with no successor is completed by ophip_frame_wait() on its ticket.  A throughput pipeline therefore keeps three frames in
 *   flight: enqueue t + 2, then wait for t.
 * ophip_frame_order_after_fine(stream): completes a kept-back fine stage of `stream` and makes `stream` wait for the fine stage of
 *   the last frame enqueued on it through this entry point (for callers that mix it with stage-by-stage calls: what they queue next on
 *   `stream` then starts behind that fine stage.  Inside the frame pipeline itself the next encoder does NOT wait for it any more --
 *   since the end of round 4 its first layer takes the CUs the fine stage's last workgroups free; OPHIP_FRAME_FINE_WAIT=1 restores the wait).
 * x3d_external (a cached keypoint encoding, read-only): the entry points that take it as a bare pointer know nothing about the stream
 *   that wrote it, so they read it on s_main only (the first layer's K / V half, which otherwise runs ahead on s_prep, stays on s_main):
 *   it must be complete with respect to s_main -- written on s_main, or behind an event s_main already waits on.  A caller that wants
 *   the K / V half on s_prep passes the cache through ophip_frame_enqueue_object with its `ready` event. */
typedef struct ophip_frame_desc {
    int B, N, M, hc, wc, hf, wf, cf;             /* cf: channels of the fine map (128) */
    int lazy_conf;                                /* 1: conf_matrix is not materialised (layout.conf = 0; result block int32 @4 = "re-run eagerly" flag) */
    int n_coarse; unsigned coarse_cross_bits;     /* bit li set: coarse layer li is a "cross" layer */
    int n_fine; unsigned fine_cross_bits; int fine_encoder_enable;
    int border_rm;
    float thr, scale_c, fine_scale;               /* scale_c = image height / hc; fine_scale = (window / 2) * image height / hf */
    double temperature;
    const float* pe;                              /* [M][256] positional-encoding table or NULL */
    const float* w_kpt;                           /* keypoint-encoder block or NULL (encoding disabled) */
    const void* w_coarse[16];                     /* packing.pack_coarse_layer_x3w8 blocks */
    const void* w_fine;                           /* packing.pack_fine_layers_bf16 block */
} ophip_frame_desc;
typedef struct ophip_frame_layout_t {
    size_t total, result_bytes;
    size_t x2d, ffcl, x3d, y3d, y2d, z3d, stats, enc_ws, conf, cws, result, i_ids, j_ids, m_bids, gt_mask, mconf, mkc, expec;
    size_t feat3d_out, feat2d_out;                /* the encoder's final rows [B][N][256] / [B][M][256] (inputs of coarse matching) */
} ophip_frame_layout_t;
int ophip_frame_layout(const ophip_frame_desc* desc, int transpose_fine, int external_x3d, ophip_frame_layout_t* layout);
int ophip_frame_enqueue(const ophip_frame_desc* desc, const ophip_frame_layout_t* layout, void* block,
                        const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                        const float* keypoints3d, long long kpts_bstride, const float* desc3d_c, long long desc_c_bstride,
                        const float* desc3d_f, long long desc_f_bstride, long long desc_f_cstride, const float* x3d_external,
                        void* host_dst, size_t host_bytes, void* s_main, void* s_prep, void* s_fine, void* s_copy, int* slot);
/* The same frame with the reference's optional inputs of padded / resized query images (see ophip_encoder_layer_x3w8_masked): query_mask
 * [B][M] bytes (1 = real cell) and / or query_scale [B][2] floats, either NULL; both must stay valid like the other inputs.  With a mask
 * the last encoder layer does not write the similarity fragments itself (the masked layer has no such form; the similarity stage derives them). */
int ophip_frame_enqueue_padded(const ophip_frame_desc* desc, const ophip_frame_layout_t* layout, void* block,
                               const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                               const float* keypoints3d, long long kpts_bstride, const float* desc3d_c, long long desc_c_bstride,
                               const float* desc3d_f, long long desc_f_bstride, long long desc_f_cstride, const float* x3d_external,
                               const unsigned char* query_mask, const float* query_scale,
                               void* host_dst, size_t host_bytes, void* s_main, void* s_prep, void* s_fine, void* s_copy, int* slot);
int ophip_frame_wait(int ticket);
int ophip_frame_order_after_fine(void* compute_stream);

/* Object cache ("kpt_encode and the first 3D self-layer are frame-invariant ... computed once per sequence and
 * cached"; the reference keeps the object block resident across a sequence's frames, OnePosePlus_inference_dataset.py:157-169, and its
 * first layer applies the layer to the 3D stream with itself as source, transformer.py:148-153, while the second layer's 2D update
 * reads only that result as K / V, :154-159).  Per object and weight set, once:
 *   x3d  [Bo][N][256]  the keypoint encoding, rows a2 + a3 (ophip_kpt_encode)
 *   y3d0 [Bo][N][256]  the first encoder layer's 3D rows                                       } ophip_encoder_object_x3w8: the launches a frame
 *   kv1  [Bo][ophip_encoder_x3w8_kv_block_bytes()]  phi(K)^T V | Ksum of those rows as layer 1's source } would run, on the 3D stream's workgroups only
 * Bo = 1 with batch strides 0 when the whole batch shares one object (BASELINE config 3), else Bo = B.  A frame enqueued with the cache
 * (ophip_frame_enqueue_object) runs layer 0 on the 2D stream alone and layer 1 with the cached rows / block; it is BIT-IDENTICAL to a
 * frame without the cache (a workgroup's arithmetic does not depend on the others of its launch; the K^T V sum adds a stream's slabs in
 * the same fixed order).  Needs n_coarse >= 2 and a first layer of kind "self" for y3d0 / kv1 (else pass them NULL: depth 1).
 * workspace of ophip_encoder_object_x3w8: ophip_encoder_x3w8_workspace_bytes(Bo, N, 1) bytes of scratch.  masked_frames: 1 = build the
 * entry for frames that carry a query_mask (they run both streams through the masked instantiation of the layer kernel, whose rounding
 * need not equal the plain one's: an entry built the same way keeps the bit-identity); a caller with both kinds of frames keeps two entries.
 * ready: a hipEvent_t recorded behind the kernels that wrote the buffers, or NULL when they are complete; ophip_frame_enqueue_object makes
 * s_main AND s_prep wait for it before their first read (the first layer's K / V half runs on s_prep). */
typedef struct ophip_object_cache {
    const float* x3d;  long long x3d_bs;          /* batch stride in floats (0: shared); may be NULL when y3d0 / kv1 are given */
    const float* y3d0; long long y3d0_bs;         /* or NULL: only the keypoint encoding is cached */
    const void* kv1;   long long kv1_bs;          /* batch stride in BYTES (0: shared); 16-byte aligned */
    void* ready;
} ophip_object_cache;
size_t ophip_encoder_x3w8_kv_block_bytes(void);
int ophip_encoder_object_x3w8(const float* x3d, int Bo, int N, const void* wpack0, const void* wpack1, void* workspace,
                              float* y3d0, void* kv1, int masked_frames, void* stream);
/* ophip_frame_enqueue_padded with the object's cache in place of x3d_external (the layout must have been made with external_x3d = 2
 * when y3d0 / kv1 are given, 1 otherwise); query_mask / query_scale may be NULL. */
int ophip_frame_enqueue_object(const ophip_frame_desc* desc, const ophip_frame_layout_t* layout, void* block,
                               const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                               const float* keypoints3d, long long kpts_bstride, const float* desc3d_c, long long desc_c_bstride,
                               const float* desc3d_f, long long desc_f_bstride, long long desc_f_cstride, const ophip_object_cache* object,
                               const unsigned char* query_mask, const float* query_scale,
                               void* host_dst, size_t host_bytes, void* s_main, void* s_prep, void* s_fine, void* s_copy, int* slot);

/* Row f-1 -- ResNetFPN_8_2 image backbone (backbone/resnet.py:20-44 BasicBlock, :85-164; called at
 * OnePosePlusModel.py:121-131) as implicit-GEMM convolutions on the bf16 matrix pipe.
 * Feature maps between layers are channels-last bf16 plane pairs (hi, lo) [B][H][W][c_pad], c_pad = channels rounded up
 * to 32 with zero padding; in plain-bf16 mode (nsplit 1) the lo pointers are unused.
 *
 * ophip_stem_conv7: conv1 7x7 stride 2 pad 3 (1 -> 128) + folded bn1 + ReLU (resnet.py:100-102,140), exact f32.
 *   image [B][1][H][W] f32; wpack = folded weights [49][128] f32 then bias [128]; out planes [B][H/2][W/2][128].
 * ophip_conv2d_bf16: one 3x3 / 1x1 convolution, stride 1 / 2, padding ks/2, no bias in the reference (BatchNorm is folded
 *   into wpack on the host: packing.pack_conv_bf16, ophip_conv_wpack_bytes bytes = hi fragments | lo fragments | bias f32).
 *   Epilogue in this order, each part optional: + bias, + residual planes (BasicBlock shortcut, resnet.py:41-43),
 *   + bilinear x2 upsampling with align_corners=True of `up` [B][Hup][Wup][cout_pad] f32 (FPN top-down, resnet.py:155-160),
 *   + `table` [Hout][Wout][cout_pad] f32 shared by all batch elements (the positional encoding of the coarse map, a1),
 *   activation (0 none, 1 ReLU, 2 LeakyReLU 0.01), then planes (out_hi/out_lo) and / or f32 channels-last
 *   out_f32 [B][Hout][Wout][out_c] (out_c <= cout_pad, multiple of 4). */
size_t ophip_conv_wpack_bytes(int cin_pad, int cout_pad, int ks);
int ophip_stem_conv7(const float* image, int B, int H, int W, const float* wpack, void* out_hi, void* out_lo, int nsplit, void* stream);
int ophip_conv2d_bf16(const void* in_hi, const void* in_lo, int B, int Hin, int Win, int cin_pad,
                      const void* wpack, int cout_pad, int ks, int stride, int act,
                      const void* res_hi, const void* res_lo, const float* up, int Hup, int Wup, const float* table,
                      void* out_hi, void* out_lo, float* out_f32, int out_c, int nsplit, void* stream);

/* Row f-3 -- the LoFTR 2D-2D matcher behind the object detector
 * (src/local_feature_object_detector/local_feature_2D_detector.py:89-144 match_worker; model
 * src/KeypointFreeSfM/loftr_for_sfm/loftr.py:16-167; its arithmetic lives in the un-vendored submodules/LoFTR and is restated from
 * the published zju3dv/LoFTR definition: parity unpinned).  Backbone and coarse encoder reuse ophip_conv2d_bf16 /
 * ophip_encoder_layer_x3w8; what is specific to the two-image matcher:
 *
 * ophip_coarse_match_2d: dual-softmax + mutual-nearest between the coarse grids of two images (loftr/utils/coarse_matching.py):
 *   sim = <f0, f1> / C / temperature exactly (nothing added to the temperature), border removal on all four sides of BOTH grids.
 *   feat0 [B][L0][256], feat1 [B][L1][256], L0 = h0c * w0c, L1 = h1c * w1c; points0 [B][L0][3] is gathered into mkpts0 [.][3] per
 *   match (the caller tabulates (x, y, 0) of cell i in image pixels = mkpts0_c); mkpts1_c = (j % w1c, j / w1c) * scale.
 * Fine stage, batched over all K matches (window W x W on both images, token rows [K][W*W][128] in HBM):
 *   ophip_fine2_gather       one image's windows: feat_cl [hf*wf][128] channels-last, centre = stride * cell(cell_ids[k]), zero padding
 *                            (loftr_module/fine_preprocess.py: F.unfold(kernel W, stride, padding W/2) + gather)
 *   ophip_rows_linear_x3     y[T][N] = act([xa | xb] W^T), split-bf16 MFMA; K = ka + kb in {128, 256}, N in {128, 256};
 *                            wpack: packing.pack_linear_x3 (ophip_rows_linear_wpack_bytes(K, N) bytes); relu != 0 -> ReLU
 *   ophip_fine2_attention    LinearAttention (loftr_module/linear_attention.py) per match, 8 heads of 16: q [K][L][128], k, v [K][S][128]
 *   ophip_rows_layernorm128  y = (residual or 0) + LayerNorm(x) * gamma + beta, rows of 128 features, eps 1e-5
 *   ophip_fine2_match        FineMatching (utils/fine_matching.py): <f0[centre], f1[r]> / sqrt(128) -> softmax -> expectation over the
 *                            normalised W x W grid, std; mkpts1_f = mkpts1_c + expectation * scale, scale = (W / 2) * (image h / fine h) */
int ophip_coarse_match_2d(const float* feat0, const float* feat1, const float* points0, long long points_bstride,
                          int B, int L0, int L1, int w0c, int w1c, double temperature, float thr, int border_rm, float scale,
                          float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                          float* mconf, float* mkpts0, float* mkpts1_c, long long* m_bids, unsigned char* gt_mask,
                          int* count, int nsplit, void* stream);
int ophip_fine2_gather(const float* feat_cl, int hf, int wf, const long long* cell_ids, int K, int wc, int stride, int W, float* out, void* stream);
/* the same over a batch of images [B][hf * wf][128]: match k reads image b_ids[k] (feat_bstride floats apart; 0: one image for every match) */
int ophip_fine2_gather_b(const float* feat_cl, long long feat_bstride, const long long* b_ids, int hf, int wf, const long long* cell_ids,
                         int K, int wc, int stride, int W, float* out, void* stream);
size_t ophip_rows_linear_wpack_bytes(int kin, int nout);
int ophip_rows_linear_x3(const float* xa, int ka, const float* xb, int kb, int T, const void* wpack, int nout, int relu, float* y, void* stream);
int ophip_fine2_attention(const float* q, const float* k, const float* v, int K, int L, int S, float* msg, void* stream);
int ophip_rows_layernorm128(const float* x, const float* gamma, const float* beta, const float* residual, int T, float* y, void* stream);
int ophip_fine2_match(const float* f0, const float* f1, const float* mkpts1_c, int K, int W, float scale, float* expec_f, float* mkpts1_f,
                      void* stream);

/* Row f-2 -- the query crop of the frame loop (local_feature_2D_detector.py:164-190 crop_img_by_bbox, called from
 * detect :208-247 and previous_pose_detect :249-280): box [x0, y0, x1, y1) of a grayscale uint8 frame [H][W] -> out [S][S]
 * float in [0, 1] (= the reference's two cv2.warpAffine passes + astype(float32) / 255: integer-shift crop, then isotropic
 * bilinear resize by S / (x1 - x0) about the crop centre, zero border, rounded to uint8 levels). */
int ophip_crop_resize_gray(const unsigned char* image, int H, int W, int x0, int y0, int x1, int y1, int S, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif

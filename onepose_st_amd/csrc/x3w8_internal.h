// Internal (not part of the C ABI): the object-cache forms of the split-bf16 coarse layer, shared by csrc/encoder_x3w8.hip (which defines
// them) and csrc/frame.hip (which issues them inside ophip_frame_enqueue_object).  See ophip_encoder_object_x3w8 in encoder_x3w8.hip.
#pragma once

// layer 0 ("self") on the 2D stream alone: the 3D stream's rows of this layer come from the cache.  kv_mode 0: with its own K / V half;
// 2: that half ran already (only_kv = true is that half alone: kv_reduce + kv_sum over the 2D stream's tiles).  Slot 0 of the workspace.
int ophip_x3w8_object_first(const float* x2d, float* y2d, int B, int L3d, int L2d, const void* wpack, const void* wpack_next, int kv_mode,
                            void* workspace, const unsigned char* mask2d, bool only_kv, void* stream);
// layer 1 on both streams: its 3D input rows (y3d0, batch stride in floats, 0 = shared) and the 3D source's K^T V | Ksum block (kv1, batch
// stride in bytes) from the cache; the 2D stream's slabs are the ones ophip_x3w8_object_first's tail left.  Slot 1 of the workspace.
int ophip_x3w8_object_second(const float* y3d0, long long y3d0_bs, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                             const void* wpack, const void* wpack_next, int is_cross, const void* kv1, long long kv1_bs,
                             void* workspace, void* frag3d, void* frag2d, const unsigned char* mask2d, void* stream);
// the plain layer (ophip_encoder_layer_x3w8{,_frag,_masked}) and its K / V half with a batch stride for the 3D rows (floats; 0 = one
// encoding shared by the batch, -1 = dense): what a cached keypoint encoding of a shared object needs
int ophip_x3w8_layer_bs(const float* x3d, long long x3d_bs, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                        const void* wpack, const void* wpack_next, int is_cross, int kv_mode, int slot,
                        void* workspace, void* frag3d, void* frag2d, const unsigned char* mask2d, bool only_kv, void* stream);

// coarse matching, eager form in the bf16 modes: true = statistics pass + a second tile pass that writes every confidence once
// (csrc/coarse_match.hip ophip_coarse_two_pass: by matrix size / OPHIP_COARSE_TWO_PASS); false = similarity store + in-place conversion pass
bool ophip_coarse_two_pass(int B, int N, int M);

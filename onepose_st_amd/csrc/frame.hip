// One call per frame: the launch sequence of onepose_st_amd/model.py's default (split-bf16) path -- rows a1-a11 of
// DESIGN.md section 1 on three streams (input kernels | encoder + coarse matching | select + fine stage) plus the read-back of the
// result block -- issued from C.  The Python loop spends ~0.4 ms per frame on ~27 ctypes calls, ~20 tensor allocations and
// ~10 stream / event operations; this entry point takes one device block (laid out by ophip_frame_layout) and returns after
// enqueueing everything (~0.1 ms).  Reference: OnePosePlusModel.py:115-203 (forward after the backbone).
// Host code only: every kernel is launched through the stage entry points of this library.
#include "tile.h"
#include "onepose_hip.h"
#include "x3w8_internal.h"
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr int kSlots = 16;          // frames that may be in flight between enqueue and wait
// The fine stage + read-back of a frame, kept back until the NEXT frame's encoder is queued (see ophip_frame_enqueue): everything
// ophip_fine_refine_bf16 and the copy need.  The caller keeps the buffers alive until ophip_frame_wait() on the frame's ticket.
struct FineJob {
    bool pending = false;
    hipStream_t s_main = nullptr, s_fine = nullptr, s_copy = nullptr;
    const float* ff = nullptr; long long fs_b = 0, fs_c = 0, fs_y = 0, fs_x = 0; int hf = 0, wf = 0;
    const float* desc_f = nullptr; long long desc_f_bs = 0, desc_f_cs = 0;
    const long long *b_ids = nullptr, *i_ids = nullptr, *j_ids = nullptr; const int* count = nullptr; int cap = 0;
    const float* mkc = nullptr; const void* w_fine = nullptr; int n_fine = 0; unsigned fine_cross_bits = 0; int fine_encoder_enable = 0;
    int wc = 0, stride = 0; float fine_scale = 0.f;
    float *expec = nullptr, *mk2d = nullptr;
    const float* qscale = nullptr;          // query_image_scale [B][2] or NULL
    void* host_dst = nullptr; const void* result_src = nullptr; size_t host_bytes = 0;
};
struct Slot {
    hipEvent_t prep_done = nullptr, coarse_done = nullptr, fine_done = nullptr, ready = nullptr, enc_done = nullptr;
    int dev = -1;
    int gen = 0;                     // how often the slot has been handed out: a ticket is gen * kSlots + index
    bool recorded = false;           // `ready` has been recorded for the current generation
    FineJob job;
};
struct DevState {
    Slot slots[kSlots];
    int next = 0;
    std::unordered_map<hipStream_t, hipEvent_t> last_fine;      // compute stream -> fine_done of the last fine stage LAUNCHED for it
    std::unordered_map<hipStream_t, int> deferred;              // compute stream -> slot whose fine stage is still kept back
};
std::mutex g_mu;                      // the tables below
std::mutex g_launch_mu;               // one thread at a time enqueues a frame or launches a kept-back fine stage (held around g_mu, never inside it)
std::unordered_map<int, DevState> g_dev;

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

int make_events(Slot& s, int dev) {
    if (s.ready && s.dev == dev) return 0;
    hipEvent_t* ev[5] = {&s.prep_done, &s.coarse_done, &s.fine_done, &s.ready, &s.enc_done};
    for (auto e : ev) {
        // `ready` is the one event the host waits on (ophip_frame_wait): a blocking wait, so that the thread that feeds the GPU sleeps instead
        // of spinning through its ~0.3 ms of slack per frame (under a cgroup CPU quota every spinning thread is quota the RANSAC pool lacks)
        hipError_t rc = hipEventCreateWithFlags(e, e == &s.ready ? (hipEventDisableTiming | hipEventBlockingSync) : hipEventDisableTiming);
        if (rc != hipSuccess) return ophip_fail(rc, "hipEventCreateWithFlags(frame)");
    }
    s.dev = dev;
    return 0;
}

}  // namespace

#define FR_CHECK(call)                                  \
    do {                                                \
        const int rc__ = (call);                        \
        if (rc__ != 0) return rc__;                     \
    } while (0)
#define FR_HIP(call, what)                              \
    do {                                                \
        const hipError_t e__ = (call);                  \
        if (e__ != hipSuccess) return ophip_fail(e__, what); \
    } while (0)

namespace {
// fine stage (a9-a11) + read-back of slot `s` on its side streams, behind `after` (an event of its compute stream later than its own
// selection; NULL: that selection's event)
int launch_fine_job(int dev, Slot& s, hipEvent_t after = nullptr) {
    FineJob& j = s.job;
    const hipStream_t on = j.s_fine;
    FR_HIP(hipStreamWaitEvent(on, after ? after : s.coarse_done, 0), "hipStreamWaitEvent(fine job)");
    if (j.qscale)
        FR_CHECK(ophip_fine_refine_bf16_scaled(j.ff, j.fs_b, j.fs_c, j.fs_y, j.fs_x, j.hf, j.wf, j.desc_f, j.desc_f_bs, j.desc_f_cs, j.b_ids, j.i_ids, j.j_ids,
                                               j.count, j.cap, j.mkc, j.w_fine, j.n_fine, j.fine_cross_bits, j.fine_encoder_enable, 3, j.wc, j.stride,
                                               j.fine_scale, j.expec, j.mk2d, nullptr, nullptr, j.qscale, on));
    else
    FR_CHECK(ophip_fine_refine_bf16(j.ff, j.fs_b, j.fs_c, j.fs_y, j.fs_x, j.hf, j.wf, j.desc_f, j.desc_f_bs, j.desc_f_cs, j.b_ids, j.i_ids, j.j_ids,
                                    j.count, j.cap, j.mkc, j.w_fine, j.n_fine, j.fine_cross_bits, j.fine_encoder_enable, 3, j.wc, j.stride,
                                    j.fine_scale, j.expec, j.mk2d, nullptr, nullptr, on));
    FR_HIP(hipEventRecord(s.fine_done, on), "hipEventRecord(fine)");
    // ---- read-back of the result block (count | b_ids | 3D points | refined 2D points) behind the fine stage ---------------------
    FR_HIP(hipStreamWaitEvent(j.s_copy, s.fine_done, 0), "hipStreamWaitEvent(fine)");
    FR_HIP(hipMemcpyAsync(j.host_dst, j.result_src, j.host_bytes, hipMemcpyDeviceToHost, j.s_copy), "hipMemcpyAsync(result block)");
    FR_HIP(hipEventRecord(s.ready, j.s_copy), "hipEventRecord(ready)");
    std::lock_guard<std::mutex> lk(g_mu);
    DevState& st = g_dev[dev];
    if (st.last_fine.size() > 64) st.last_fine.clear();
    st.last_fine[j.s_main] = s.fine_done;                   // published only once it is recorded: the next encoder on s_main orders behind it
    auto it = st.deferred.find(j.s_main);
    if (it != st.deferred.end() && &st.slots[it->second] == &s) st.deferred.erase(it);
    j.pending = false;
    s.recorded = true;
    return 0;
}

// OPHIP_FRAME_DEFER_FINE=0: the fine stage follows its own frame's selection at once (the round-2 order)
// OPHIP_FRAME_KV_FIRST: where the first encoder layer's K / V half is issued (see frame_enqueue_impl): "prep" (1), "main" (2), "off" (0).
// Default: "prep" -- behind the input kernels on their stream, beside whatever runs when they run.  (A padded frame's half, and the half
// of a frame whose cached encoding came in as a bare pointer, stay on the compute stream whatever this says.)
bool fine_on_main_enabled();
int kv_first_mode() {
    static const int mode = [] {
        const char* e = getenv("OPHIP_FRAME_KV_FIRST");
        if (!e || !e[0]) return 1;          // (round 4's default was "main" while the encoder still waited for the previous fine stage)
        if (e[0] == 'p' || e[0] == '1') return 1;
        if (e[0] == 'm' || e[0] == '2') return 2;
        return 0;
    }();
    return mode;
}

// OPHIP_FRAME_FINE_ON_MAIN=1: the kept-back fine stage on the compute stream, conf / selection on the side stream (experiment, see
// ophip_frame_enqueue_padded; default 0 = round 3's assignment, which measured 1-2 % faster again in round 4)
bool fine_on_main_enabled() {
    static const bool on = [] { const char* e = getenv("OPHIP_FRAME_FINE_ON_MAIN"); return e && e[0] == '1'; }();
    return on;
}

// OPHIP_FRAME_FINE_WAIT=1: the encoder waits for the previous frame's fine stage (rounds 2-4: "attn_apply never shares the chip").  Default
// since the end of round 4: no wait.  The fine stage's 1 400 workgroups are 2.7 rounds on 512 slots, so the last quarter of its time runs on a
// thinning set of CUs; the encoder's first layer now fills them instead of starting when the last workgroup has ended (interleaved A/B on
// one box, c2: 1 344 -> 1 366 frames/s at 20 steps, 1 429 -> 1 458 at 100; with the first layer's K / V half on the input stream, below,
// 1 391 and 1 496).
bool fine_wait_enabled() {
    static const bool on = [] { const char* e = getenv("OPHIP_FRAME_FINE_WAIT"); return e && e[0] == '1'; }();
    return on;
}
bool defer_fine_enabled() {
    static const bool on = [] { const char* e = getenv("OPHIP_FRAME_DEFER_FINE"); return !(e && e[0] == '0'); }();
    return on;
}
}  // namespace

extern "C" int ophip_frame_layout(const ophip_frame_desc* d, int transpose_fine, int external_x3d, ophip_frame_layout_t* L) {
    if (!d || !L) return ophip_bad_arg(__func__, "null pointer");
    if (d->B < 1 || d->N < 1 || d->M < 1 || d->hc * d->wc != d->M || d->n_coarse < 1 || d->n_coarse > 16 || d->cf < 1)
        return ophip_bad_arg(__func__, "bad sizes");
    const size_t B = d->B, N = d->N, M = d->M, cap = B * N, C = 256;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = up256(o + bytes); return at; };
    L->x2d = take(B * M * C * 4);
    L->ffcl = transpose_fine ? take(B * (size_t)d->hf * d->wf * d->cf * 4) : 0;
    if (external_x3d < 0 || external_x3d > 2) return ophip_bad_arg(__func__, "external_x3d must be 0, 1 or 2");
    if (external_x3d == 2 && (d->n_coarse < 2 || (d->coarse_cross_bits & 1u)))
        return ophip_bad_arg(__func__, "a cached first layer needs n_coarse >= 2 and a first layer of kind \"self\"");
    L->x3d = external_x3d ? 0 : take(B * N * C * 4);
    L->y3d = take(B * N * C * 4);
    L->y2d = take(B * M * C * 4);
    L->z3d = external_x3d ? take(B * N * C * 4) : 0;
    L->stats = take((4 * B + 4) * 4);
    L->enc_ws = take(ophip_encoder_x3w8_workspace_bytes(d->B, d->N, d->M));
    L->conf = d->lazy_conf ? 0 : take(B * N * M * 4);          // lazy form: conf_matrix is never stored
    L->cws = take(ophip_coarse_workspace_floats(d->B, d->N, d->M) * 4);
    L->result = take(16 + 28 * cap);
    L->i_ids = take(cap * 8);
    L->j_ids = take(cap * 8);
    L->m_bids = take(cap * 8);
    L->gt_mask = take(cap);
    L->mconf = take(cap * 4);
    L->mkc = take(cap * 8);
    L->expec = take(cap * 12);
    L->total = o;
    L->result_bytes = 16 + 28 * cap;
    // where the encoder's final rows end up (the ping-pong of ophip_frame_enqueue): what a caller needs to materialise a lazy conf_matrix
    // (plain / cached encoding: layer 0 writes y3d, layer 1 z3d (resp. back into x3d), ...; cached first layer: layer 1 writes y3d, layer 2 z3d, ...)
    const size_t z3 = external_x3d ? L->z3d : L->x3d;
    if (external_x3d == 2) L->feat3d_out = (d->n_coarse % 2 == 0) ? L->y3d : z3;
    else L->feat3d_out = (d->n_coarse % 2 == 0) ? z3 : L->y3d;
    L->feat2d_out = (d->n_coarse % 2 == 0) ? L->x2d : L->y2d;
    return 0;
}

namespace {
// obj: the object's cache or NULL.  ext_main_only: the cache came in as a bare x3d_external pointer (ophip_frame_enqueue{,_padded}): nothing is
// known about the stream that wrote it, so it is read on s_main only (see the header).
int frame_enqueue_impl(const ophip_frame_desc* d, const ophip_frame_layout_t* L, void* blob_,
                       const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                       const float* kpts, long long kpts_bs, const float* desc_c, long long desc_c_bs,
                       const float* desc_f, long long desc_f_bs, long long desc_f_cs, const ophip_object_cache* obj, bool ext_main_only,
                       const unsigned char* qmask, const float* qscale,
                       void* host_dst, size_t host_bytes, void* s_main_, void* s_prep_, void* s_fine_, void* s_copy_,
                       int* slot_out) {
    if (!d || !L || !blob_ || !feat_c || !feat_f || !kpts || !desc_c || !desc_f || !host_dst || !s_fine_ || !s_copy_ || !slot_out)
        return ophip_bad_arg(__func__, "null pointer");
    if (host_bytes < 16 || host_bytes > L->result_bytes) return ophip_bad_arg(__func__, "host_bytes");
    const bool deep = obj && obj->y3d0;                      // the first layer's 3D rows and layer 1's 3D-source block come from the cache
    const float* x3d_external = (obj && !deep) ? obj->x3d : nullptr;
    const long long x3d_ext_bs = obj ? obj->x3d_bs : -1;
    if (obj && !deep && !obj->x3d) return ophip_bad_arg(__func__, "object cache without buffers");
    if (deep && (!obj->kv1 || (reinterpret_cast<uintptr_t>(obj->kv1) & 15))) return ophip_bad_arg(__func__, "object cache: kv1 missing or not 16-byte aligned");
    if (deep && (d->n_coarse < 2 || (d->coarse_cross_bits & 1u))) return ophip_bad_arg(__func__, "a cached first layer needs n_coarse >= 2 and a first layer of kind \"self\"");
    if ((L->x3d == 0) != (obj != nullptr)) return ophip_bad_arg(__func__, "x3d_external / object cache does not match the layout");
    {   // which of the two external layouts the block was laid out for (they differ in where the final 3D rows end up)
        const size_t want = deep ? ((d->n_coarse % 2 == 0) ? L->y3d : L->z3d) : ((d->n_coarse % 2 == 0) ? (obj ? L->z3d : L->x3d) : L->y3d);
        if (L->feat3d_out != want) return ophip_bad_arg(__func__, "layout was made for another external_x3d mode");
    }
    char* blob = static_cast<char*>(blob_);
    hipStream_t s_main = (hipStream_t)s_main_, s_prep = (hipStream_t)s_prep_, s_fine = (hipStream_t)s_fine_, s_copy = (hipStream_t)s_copy_;
    const int B = d->B, N = d->N, M = d->M, cap = B * N;
    int dev = 0;
    FR_HIP(hipGetDevice(&dev), "hipGetDevice");
    // Order of a pipeline's frames on the chip (defer mode, the default; the caller passes distinct side streams):
    //     s_main:  encoder(t+1) | similarity(t+1) | statistics merge, confidence, selection (t+1) ...... | encoder(t+2) ...
    //     s_fine:                                 | fine stage(t)  ------------------------------------>|
    // Frame t's fine stage is kept back until frame t + 1's encoder and similarity tiles are queued and runs BESIDE the HBM-bound half
    // of frame t + 1's coarse matching (conf_kernel streams 269 MB, with non-temporal accesses so that it does not evict the fine
    // stage's weights from L2, and needs no matrix pipe; the fine stage is matrix-bound and moves little).  The encoder's 150 KB workgroups
    // leave no LDS for a second kernel on a CU; since the end of round 4 frame t + 2's encoder no longer waits for frame t's fine stage to
    // END (fine_wait_enabled()): its first layer takes the CUs that stage's last, thinning round of workgroups frees one by one.
    // A frame with no successor is completed by ophip_frame_wait() (or ophip_frame_order_after_fine()).
    const bool defer = defer_fine_enabled() && s_fine != s_main;
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    Slot* slot;
    Slot* kept = nullptr;             // the previous frame of this compute stream whose fine stage is still kept back
    int idx, gen;
    hipEvent_t prev_fine = nullptr, prev_ready = nullptr;
    bool stale_job = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DevState& st = g_dev[dev];
        idx = st.next;
        st.next = (st.next + 1) % kSlots;
        slot = &st.slots[idx];
        FR_CHECK(make_events(*slot, dev));
        stale_job = slot->job.pending;
    }
    // (a slot that comes round with its fine stage still kept back -- its stream saw no further frame --: complete that frame first)
    if (stale_job) FR_CHECK(launch_fine_job(dev, *slot));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DevState& st = g_dev[dev];
        if (slot->recorded) prev_ready = slot->ready;
        auto it = st.last_fine.find(s_main);
        if (it != st.last_fine.end()) prev_fine = it->second;
        auto dt = st.deferred.find(s_main);
        if (dt != st.deferred.end()) kept = &st.slots[dt->second];
    }
    // a slot comes round again after kSlots frames: its previous frame must have left the GPU before its events are re-recorded
    // (normally long done -- a pipeline keeps a few frames in flight --, then this returns at once)
    if (prev_ready) FR_HIP(hipEventSynchronize(prev_ready), "hipEventSynchronize(slot reuse)");
    {
        std::lock_guard<std::mutex> lk(g_mu);
        slot->recorded = false;
        gen = ++slot->gen;
    }
    // From here on the slot belongs to this frame.  If any of the ~20 launches below fails, the frame is abandoned half queued: the guard
    // then waits until nothing of it is left on the four streams (so no event of the slot is pending and no kernel still reads the
    // caller's block) and hands the slot back clean -- not recorded, no kept-back fine stage -- before the error is returned; the previous
    // frame of the stream keeps its own kept-back fine stage (ophip_frame_wait / the next enqueue launch it as usual).
    struct AbandonGuard {
        Slot* slot; hipStream_t st[4]; bool armed = true;
        ~AbandonGuard() {
            if (!armed) return;
            for (hipStream_t s : st) if (s) (void)hipStreamSynchronize(s);
            std::lock_guard<std::mutex> lk(g_mu);
            slot->recorded = false;
            slot->job.pending = false;
        }
    } abandon{slot, {s_main, s_prep, s_fine, s_copy}};
    auto F = [&](size_t off) { return reinterpret_cast<float*>(blob + off); };
    auto I64 = [&](size_t off) { return reinterpret_cast<long long*>(blob + off); };

    // ---- input kernels (a1-a3 + the fine map's transpose): on their own stream when the caller's tensors are complete ----------
    hipStream_t sin = s_prep ? s_prep : s_main;
    OphipRange frame_range("ophip_frame_enqueue");
    struct StageRange {                 // one open roctx range at a time, closed on every exit path
        bool open = false;
        void next(const char* name) { if (open) ophip_range_pop(); ophip_range_push(name); open = true; }
        ~StageRange() { if (open) ophip_range_pop(); }
    } stage;
    stage.next("a1-a3 input kernels");
    // OPHIP_FRAME_PREP_AFTER_SIM=1 (experiment): this frame's input kernels start when the previous frame's similarity tiles have ended
    // (beside its confidence pass and the fine stage) instead of whenever the host queued them (beside its last encoder layers)
    static const bool prep_after_sim = [] { const char* e = getenv("OPHIP_FRAME_PREP_AFTER_SIM"); return e && e[0] == '1'; }();
    if (prep_after_sim && s_prep && kept) FR_HIP(hipStreamWaitEvent(s_prep, kept->enc_done, 0), "hipStreamWaitEvent(previous similarity)");
    float* x2d = F(L->x2d);
    FR_CHECK(ophip_pe_add_transpose(feat_c, d->pe, x2d, B, 256, M, sin));
    const float* ff = feat_f;
    if (L->ffcl) {                                          // NCHW fine map -> channels-last rows (fs_* then describe the copy)
        FR_CHECK(ophip_transpose_cl(feat_f, F(L->ffcl), B, d->cf, d->hf * d->wf, sin));
        ff = F(L->ffcl);
        fs_b = (long long)d->hf * d->wf * d->cf; fs_c = 1; fs_y = (long long)d->wf * d->cf; fs_x = d->cf;
    }
    if (obj && obj->ready) {                                // the cache's buffers: complete before either stream's first read of them
        FR_HIP(hipStreamWaitEvent(s_main, (hipEvent_t)obj->ready, 0), "hipStreamWaitEvent(object cache)");
        if (s_prep) FR_HIP(hipStreamWaitEvent(s_prep, (hipEvent_t)obj->ready, 0), "hipStreamWaitEvent(object cache, input stream)");
    }
    float* x3d = x3d_external ? const_cast<float*>(x3d_external) : (deep ? nullptr : F(L->x3d));
    if (!obj) {
        if (d->w_kpt) FR_CHECK(ophip_kpt_encode(kpts, kpts_bs, desc_c, desc_c_bs, d->w_kpt, F(L->stats), x3d, B, N, sin));
        else {
            if (B > 1 && desc_c_bs == 0) return ophip_bad_arg(__func__, "shared descriptors need the keypoint encoder");
            FR_CHECK(ophip_transpose_cl(desc_c, x3d, B, 256, N, sin));
        }
    }
    stage.next("a4-a6 coarse encoder");
    // The first encoder layer's K / V half (kv_reduce + kv_sum: ~20 us of the frame at c2) reads this frame's input rows only.  It used to
    // be the first thing BEHIND the wait for the previous frame's fine stage, i.e. on the critical path of every frame; it now goes out
    // ahead of that wait: "prep" = behind the input kernels on their stream (beside whatever runs when they run), "main" = on the
    // compute stream in front of the wait (in the tail of the previous fine stage), "off" = inside the layer call as before.
    const int kv_first = kv_first_mode();
    const bool kv_hoisted = kv_first != 0 && d->n_coarse > 0;
    // (a padded frame's cell mask is made by the caller on the compute stream just before this call: its K / V half stays on that stream)
    // (a cached encoding that came in as a bare pointer is read on the compute stream only: ext_main_only; with a cached first layer the
    //  half covers the 2D stream alone and reads nothing of the cache)
    const bool kv_on_prep = kv_first == 1 && s_prep && !qmask && !(x3d_external && ext_main_only);
    auto kv_first_half = [&](hipStream_t on) -> int {
        if (deep) return ophip_x3w8_object_first(x2d, nullptr, B, N, M, d->w_coarse[0], nullptr, 0, blob + L->enc_ws, qmask, true, on);
        if (x3d_external) return ophip_x3w8_layer_bs(x3d, x3d_ext_bs, x2d, nullptr, nullptr, B, N, M, d->w_coarse[0], nullptr, 0, 0, 0, blob + L->enc_ws,
                                                     nullptr, nullptr, qmask, true, on);
        return ophip_encoder_kv_first_x3w8(x3d, x2d, B, N, M, d->w_coarse[0], 0, blob + L->enc_ws, qmask, on);
    };
    if (kv_hoisted && kv_on_prep) FR_CHECK(kv_first_half(s_prep));
    if (s_prep) {
        FR_HIP(hipEventRecord(slot->prep_done, s_prep), "hipEventRecord(prep)");
        FR_HIP(hipStreamWaitEvent(s_main, slot->prep_done, 0), "hipStreamWaitEvent(prep)");
    }
    if (kv_hoisted && !kv_on_prep) FR_CHECK(kv_first_half(s_main));
    // ---- a4-a6: coarse encoder (OPHIP_FRAME_FINE_WAIT=1: behind the previous frame's fine stage) ----------------------------------
    if (prev_fine && fine_wait_enabled()) FR_HIP(hipStreamWaitEvent(s_main, prev_fine, 0), "hipStreamWaitEvent(previous fine)");      // (a no-op when it ran on s_main)
    float *y3d = F(L->y3d), *y2d = F(L->y2d), *y2 = y2d, *x2 = x2d;
    float* z3d = obj ? F(L->z3d) : x3d;                   // a cached encoding is read-only: ping-pong between y and z
    float *x3 = x3d, *y3 = y3d;
    // the last layer also writes its output as the similarity kernel's operand fragments (into the coarse workspace): no
    // frag_planes launch in front of the similarity tiles
    void *planes3d = nullptr, *planes2d = nullptr;
    FR_CHECK(ophip_coarse_frag_planes(F(L->cws), B, N, M, &planes3d, &planes2d));
    for (int li = 0; li < d->n_coarse; ++li) {
        const void* nxt = li + 1 < d->n_coarse ? d->w_coarse[li + 1] : nullptr;
        const int is_cross = (d->coarse_cross_bits >> li) & 1;
        const int kv_mode = li > 0 ? 1 : (kv_hoisted ? 2 : 0);
        const bool last = li + 1 == d->n_coarse;
        void* f3 = (last && !qmask) ? planes3d : nullptr;      // (the masked layer has no fragment-writing form: the similarity stage derives them)
        void* f2 = (last && !qmask) ? planes2d : nullptr;
        if (deep && li == 0) {
            // layer 0 on the 2D stream alone; the 3D stream's rows of this layer are the cache's
            FR_CHECK(ophip_x3w8_object_first(x2, y2, B, N, M, d->w_coarse[0], nxt, kv_mode, blob + L->enc_ws, qmask, false, s_main));
            x3 = const_cast<float*>(obj->y3d0);
            y3 = y3d;
            float* t2 = x2; x2 = y2; y2 = t2;
            continue;
        }
        if (deep && li == 1) {
            FR_CHECK(ophip_x3w8_object_second(x3, obj->y3d0_bs, x2, y3, y2, B, N, M, d->w_coarse[1], nxt, is_cross, obj->kv1, obj->kv1_bs,
                                              blob + L->enc_ws, f3, f2, qmask, s_main));
            x3 = y3; y3 = z3d;
            float* t2 = x2; x2 = y2; y2 = t2;
            continue;
        }
        if (x3d_external && li == 0)      // the cached encoding with its own batch stride (0: one object shared by the batch)
            FR_CHECK(ophip_x3w8_layer_bs(x3, x3d_ext_bs, x2, y3, y2, B, N, M, d->w_coarse[li], nxt, is_cross, kv_mode, 0, blob + L->enc_ws, f3, f2, qmask, false, s_main));
        else if (qmask)      // padded query cells (query_image_mask): the masked layer; the similarity stage then derives its operand fragments itself
            FR_CHECK(ophip_encoder_layer_x3w8_masked(x3, x2, y3, y2, B, N, M, d->w_coarse[li], nxt, is_cross, kv_mode, li & 1,
                                                     blob + L->enc_ws, qmask, s_main));
        else if (last)
            FR_CHECK(ophip_encoder_layer_x3w8_frag(x3, x2, y3, y2, B, N, M, d->w_coarse[li], nxt, is_cross, kv_mode, li & 1,
                                                   blob + L->enc_ws, planes3d, planes2d, s_main));
        else
            FR_CHECK(ophip_encoder_layer_x3w8(x3, x2, y3, y2, B, N, M, d->w_coarse[li], nxt, is_cross, kv_mode, li & 1, blob + L->enc_ws, s_main));
        float* nx3 = y3;
        y3 = li == 0 ? z3d : x3;
        x3 = nx3;
        float* t2 = x2; x2 = y2; y2 = t2;
    }
    // ---- a7 + a8: coarse matching; selection, fine stage and read-back on their side streams -----------------------------------
    stage.next("a7-a11 coarse matching, fine stage, read-back");
    float* conf = d->lazy_conf ? nullptr : F(L->conf);
    float* cws = F(L->cws);
    int* count = reinterpret_cast<int*>(blob + L->result);
    long long* b_ids = I64(L->result + 16);
    float* mk3d = F(L->result + 16 + 8 * (size_t)cap);
    float* mk2d = F(L->result + 16 + 20 * (size_t)cap);
    unsigned char* gt_mask = reinterpret_cast<unsigned char*>(blob + L->gt_mask);
    const int nsplit_flags = qmask ? 3 : (3 | OPHIP_COARSE_PLANES_READY);
    // (the eager form's two-pass variant -- large N x M, ophip_coarse_two_pass -- has a second matrix-bound tile pass like the lazy form:
    //  both tile passes stay on the compute stream in front of the kept-back fine stage)
    const bool tiles_twice = d->lazy_conf || ophip_coarse_two_pass(B, N, M);
    if (!defer)
    FR_CHECK(ophip_coarse_match_masked(x3, x2, kpts, kpts_bs, B, N, M, d->wc, d->temperature, d->thr, d->border_rm, d->scale_c, conf, cws,
                                       b_ids, I64(L->i_ids), I64(L->j_ids), F(L->mconf), mk3d, F(L->mkc), I64(L->m_bids), gt_mask, count,
                                       nsplit_flags, 1, qmask, qscale, s_main));
    // With the input kernels on their own stream nothing is left on the compute stream that could run beside the fine stage
    // (the next encoder waits for it anyway), so selection + fine stage stay in order on the compute stream: a dependent kernel
    // on the same queue starts ~2 us after its producer, one behind a cross-stream event 10-17 us after (rocprof trace).
    // (1265 against 1239 frames/s over three runs each at c2)
    if (s_prep && !defer) s_fine = s_main;
    if (defer) {
        // The similarity tiles first, alone: matrix-bound like the fine stage, and two matrix-bound kernels side by side only stretch
        // each other (292 + 291 us together against 80 + 236 us apart, rocprof trace).  Then the previous frame's fine stage on the side
        // stream, BESIDE this frame's HBM-bound half (statistics merge, conf_kernel, selection: ~120 us that need no matrix pipe).
        // (Fine stage on the compute stream and the HBM-bound half on the side stream instead: the same within noise.)
        FR_CHECK(ophip_coarse_match_masked(x3, x2, kpts, kpts_bs, B, N, M, d->wc, d->temperature, d->thr, d->border_rm, d->scale_c, conf, cws,
                                           b_ids, I64(L->i_ids), I64(L->j_ids), F(L->mconf), mk3d, F(L->mkc), I64(L->m_bids), gt_mask, count,
                                           nsplit_flags, tiles_twice ? 1 : 4, qmask, qscale, s_main));      // (lazy / two-pass form: the second tile pass is matrix-bound too)
        FR_HIP(hipEventRecord(slot->enc_done, s_main), "hipEventRecord(similarity)");
        // Which of the two halves that follow the similarity tiles stays on the compute stream (OPHIP_FRAME_FINE_ON_MAIN, default 0):
        //   0: this frame's HBM-bound half (statistics merge, conf, selection) does; the previous frame's fine stage runs on the side stream.
        //   1: the fine stage does (the next encoder then follows it on the same hardware queue: ~2 us instead of the 10-17 us of a
        //      cross-stream event) and the HBM-bound half goes to the side stream.  Measured in round 3 (equal) and again in round 4 with
        //      the two-kernel selection: 1 362 against 1 391 frames/s, three interleaved 100-step runs each -- the side stream's kernels
        //      start 10-17 us later behind THEIR cross-stream edge and the fine stage, first in its queue, takes the chip before them.
        const bool fine_on_main = fine_on_main_enabled() && !tiles_twice;
        const hipStream_t s_tail = fine_on_main ? s_fine : s_main;          // where this frame's merge / conf / selection run
        if (fine_on_main) FR_HIP(hipStreamWaitEvent(s_tail, slot->enc_done, 0), "hipStreamWaitEvent(similarity)");
        FR_CHECK(ophip_coarse_match_masked(x3, x2, kpts, kpts_bs, B, N, M, d->wc, d->temperature, d->thr, d->border_rm, d->scale_c, conf, cws,
                                           b_ids, I64(L->i_ids), I64(L->j_ids), F(L->mconf), mk3d, F(L->mkc), I64(L->m_bids), gt_mask, count,
                                           nsplit_flags, tiles_twice ? 2 : (8 | 2), qmask, qscale, s_tail));
        FR_HIP(hipEventRecord(slot->coarse_done, s_tail), "hipEventRecord(coarse)");
        // the kept-back fine stage of the previous frame, behind this frame's similarity tiles and beside the rest.  Submitted AFTER this
        // frame's confidence pass and selection, so that those are in their hardware queue first (HIP maps streams onto a few hardware
        // queues; the read-back stream's wait for that fine stage may share one with another stream of the frame).
        // (Frames alternating between the two streams -- t + 2 right behind fine(t) on one queue -- was built and measured in round 3:
        //  1 429 against 1 470 frames/s; the read-back's packets then sat between fine(t) and encoder(t + 2).)
        if (kept) FR_CHECK(launch_fine_job(dev, *kept, kept->job.s_fine == s_main ? nullptr : slot->enc_done));      // (on the compute stream: behind the tiles by stream order, and behind its own selection's event)
        if (fine_on_main) s_fine = s_main;                                   // this frame's own fine stage (launched by its successor) runs on the compute stream
        FineJob& j = slot->job;
        j.s_main = s_main; j.s_fine = s_fine; j.s_copy = s_copy;
        j.ff = ff; j.fs_b = fs_b; j.fs_c = fs_c; j.fs_y = fs_y; j.fs_x = fs_x; j.hf = d->hf; j.wf = d->wf;
        j.desc_f = desc_f; j.desc_f_bs = desc_f_bs; j.desc_f_cs = desc_f_cs;
        j.b_ids = b_ids; j.i_ids = I64(L->i_ids); j.j_ids = I64(L->j_ids); j.count = count; j.cap = cap;
        j.mkc = F(L->mkc); j.w_fine = d->w_fine; j.n_fine = d->n_fine; j.fine_cross_bits = d->fine_cross_bits; j.fine_encoder_enable = d->fine_encoder_enable;
        j.wc = d->wc; j.stride = d->hf / d->hc; j.fine_scale = d->fine_scale;
        j.expec = F(L->expec); j.mk2d = mk2d;
        j.qscale = qscale;
        j.host_dst = host_dst; j.result_src = blob + L->result; j.host_bytes = host_bytes;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            j.pending = true;
            g_dev[dev].deferred[s_main] = idx;
        }
        abandon.armed = false;
        *slot_out = gen * kSlots + idx;
        return 0;
    }
    if (s_fine != s_main) {
        FR_HIP(hipEventRecord(slot->coarse_done, s_main), "hipEventRecord(coarse)");
        FR_HIP(hipStreamWaitEvent(s_fine, slot->coarse_done, 0), "hipStreamWaitEvent(coarse)");
    }
    FR_CHECK(ophip_coarse_match_masked(x3, x2, kpts, kpts_bs, B, N, M, d->wc, d->temperature, d->thr, d->border_rm, d->scale_c, conf, cws,
                                       b_ids, I64(L->i_ids), I64(L->j_ids), F(L->mconf), mk3d, F(L->mkc), I64(L->m_bids), gt_mask, count, 3, 2, qmask, qscale, s_fine));
    // ---- a9-a11: fine refinement (grid sized by capacity, device-side count) ----------------------------------------------------
    if (qscale)
        FR_CHECK(ophip_fine_refine_bf16_scaled(ff, fs_b, fs_c, fs_y, fs_x, d->hf, d->wf, desc_f, desc_f_bs, desc_f_cs, b_ids, I64(L->i_ids), I64(L->j_ids), count, cap,
                                               F(L->mkc), d->w_fine, d->n_fine, d->fine_cross_bits, d->fine_encoder_enable, 3, d->wc, d->hf / d->hc, d->fine_scale,
                                               F(L->expec), mk2d, nullptr, nullptr, qscale, s_fine));
    else
    FR_CHECK(ophip_fine_refine_bf16(ff, fs_b, fs_c, fs_y, fs_x, d->hf, d->wf, desc_f, desc_f_bs, desc_f_cs, b_ids, I64(L->i_ids), I64(L->j_ids), count, cap,
                                    F(L->mkc), d->w_fine, d->n_fine, d->fine_cross_bits, d->fine_encoder_enable, 3, d->wc, d->hf / d->hc, d->fine_scale,
                                    F(L->expec), mk2d, nullptr, nullptr, s_fine));
    FR_HIP(hipEventRecord(slot->fine_done, s_fine), "hipEventRecord(fine)");
    {                                                       // published only once it is recorded: the next frame on s_main orders behind it
        std::lock_guard<std::mutex> lk(g_mu);
        DevState& st = g_dev[dev];
        if (st.last_fine.size() > 64) st.last_fine.clear();
        st.last_fine[s_main] = slot->fine_done;
    }
    // ---- read-back of the result block (count | b_ids | 3D points | refined 2D points) behind the fine stage ---------------------
    FR_HIP(hipStreamWaitEvent(s_copy, slot->fine_done, 0), "hipStreamWaitEvent(fine)");
    FR_HIP(hipMemcpyAsync(host_dst, blob + L->result, host_bytes, hipMemcpyDeviceToHost, s_copy), "hipMemcpyAsync(result block)");
    FR_HIP(hipEventRecord(slot->ready, s_copy), "hipEventRecord(ready)");
    {
        std::lock_guard<std::mutex> lk(g_mu);
        slot->recorded = true;
    }
    abandon.armed = false;
    *slot_out = gen * kSlots + idx;                         // ticket: ophip_frame_wait rejects nothing but knows a reused slot's frame is done
    return 0;
}

}  // namespace

extern "C" int ophip_frame_enqueue_padded(const ophip_frame_desc* d, const ophip_frame_layout_t* L, void* blob,
                                          const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                                          const float* kpts, long long kpts_bs, const float* desc_c, long long desc_c_bs,
                                          const float* desc_f, long long desc_f_bs, long long desc_f_cs, const float* x3d_external,
                                          const unsigned char* qmask, const float* qscale,
                                          void* host_dst, size_t host_bytes, void* s_main, void* s_prep, void* s_fine, void* s_copy, int* slot_out) {
    ophip_object_cache oc{};
    oc.x3d = x3d_external; oc.x3d_bs = -1;                 // dense [B][N][256]
    return frame_enqueue_impl(d, L, blob, feat_c, feat_f, fs_b, fs_c, fs_y, fs_x, kpts, kpts_bs, desc_c, desc_c_bs, desc_f, desc_f_bs, desc_f_cs,
                              x3d_external ? &oc : nullptr, true, qmask, qscale, host_dst, host_bytes, s_main, s_prep, s_fine, s_copy, slot_out);
}

extern "C" int ophip_frame_enqueue(const ophip_frame_desc* d, const ophip_frame_layout_t* L, void* blob,
                                   const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                                   const float* kpts, long long kpts_bs, const float* desc_c, long long desc_c_bs,
                                   const float* desc_f, long long desc_f_bs, long long desc_f_cs, const float* x3d_external,
                                   void* host_dst, size_t host_bytes, void* s_main, void* s_prep, void* s_fine, void* s_copy, int* slot_out) {
    return ophip_frame_enqueue_padded(d, L, blob, feat_c, feat_f, fs_b, fs_c, fs_y, fs_x, kpts, kpts_bs, desc_c, desc_c_bs, desc_f, desc_f_bs, desc_f_cs,
                                      x3d_external, nullptr, nullptr, host_dst, host_bytes, s_main, s_prep, s_fine, s_copy, slot_out);
}

extern "C" int ophip_frame_enqueue_object(const ophip_frame_desc* d, const ophip_frame_layout_t* L, void* blob,
                                          const float* feat_c, const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x,
                                          const float* kpts, long long kpts_bs, const float* desc_c, long long desc_c_bs,
                                          const float* desc_f, long long desc_f_bs, long long desc_f_cs, const ophip_object_cache* object,
                                          const unsigned char* qmask, const float* qscale,
                                          void* host_dst, size_t host_bytes, void* s_main, void* s_prep, void* s_fine, void* s_copy, int* slot_out) {
    if (object && !object->x3d && !object->y3d0) return ophip_bad_arg(__func__, "object cache without buffers");
    return frame_enqueue_impl(d, L, blob, feat_c, feat_f, fs_b, fs_c, fs_y, fs_x, kpts, kpts_bs, desc_c, desc_c_bs, desc_f, desc_f_bs, desc_f_cs,
                              object, false, qmask, qscale, host_dst, host_bytes, s_main, s_prep, s_fine, s_copy, slot_out);
}

extern "C" int ophip_frame_wait(int ticket) {
    if (ticket < kSlots) return ophip_bad_arg(__func__, "not a ticket of ophip_frame_enqueue");
    const int slot = ticket % kSlots, gen = ticket / kSlots;
    int dev = 0;
    FR_HIP(hipGetDevice(&dev), "hipGetDevice");
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> launch_lock(g_launch_mu);
        Slot* kept = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            Slot& s = g_dev[dev].slots[slot];
            if (s.gen > gen) return 0;                      // the slot was handed out again: enqueue waited for this frame before that
            if (s.gen == gen && s.job.pending) kept = &s;   // no later frame on its stream yet: its fine stage goes out now
        }
        if (kept) FR_CHECK(launch_fine_job(dev, *kept));
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const Slot& s = g_dev[dev].slots[slot];
        if (s.gen > gen) return 0;
        if (s.gen < gen || !s.recorded) return ophip_bad_arg(__func__, "ticket of a frame that was never (completely) enqueued on this device");
        ev = s.ready;
    }
    FR_HIP(hipEventSynchronize(ev), "hipEventSynchronize(frame)");
    return 0;
}

extern "C" int ophip_frame_order_after_fine(void* compute_stream) {
    int dev = 0;
    FR_HIP(hipGetDevice(&dev), "hipGetDevice");
    hipEvent_t ev = nullptr;
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    Slot* kept = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DevState& st = g_dev[dev];
        auto dt = st.deferred.find((hipStream_t)compute_stream);
        if (dt != st.deferred.end()) kept = &st.slots[dt->second];
    }
    if (kept) FR_CHECK(launch_fine_job(dev, *kept));      // a kept-back fine stage goes out first
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& m = g_dev[dev].last_fine;
        auto it = m.find((hipStream_t)compute_stream);
        if (it != m.end()) { ev = it->second; m.erase(it); }
    }
    if (ev) FR_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, ev, 0), "hipStreamWaitEvent(fine)");
    return 0;
}

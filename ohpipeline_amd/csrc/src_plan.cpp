// src_plan.cpp -- host-side planning for the block resampler kernel: group output messages into
// contiguous stream segments, cut the segments into phase-aligned blocks, hand everything that is not a whole
// block (segment heads/tails, segments that break an alignment rule) to the generic kernel as message pieces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <utility>
#include <vector>

#include "ohgpu_internal.h"
#include "src_block_common.h"

namespace ohgpu {

void free_src_fast(ohgpu_ctx* ctx, ohgpu_batch* b)
{
    SrcFastPlan& f = b->fast;
    // (every d_* below points into the slab, and the slab goes back to the context's cache of device blocks for the next batch to
    // write into: nothing of this one may still be running -- the planes' fill, the last launch)
    if (f.planes_ready) { (void)hipEventSynchronize(f.planes_ready); (void)hipEventDestroy(f.planes_ready); }
    if (f.d_slab) (void)batch_wait_last_launch(b);
    if (f.d_slab) { if (ctx) ctx_dev_free(ctx, f.d_slab); else (void)hipFree(f.d_slab); }
    f = SrcFastPlan();
}

// A plan's device arrays live in ONE allocation, filled by ONE copy: at a live pipeline's cadence (a batch per 5 ms period,
// ohgpu_src_process_host) the seven allocations, seven synchronous copies and seven frees of round 1 were most of the call.
struct Slab {
    std::vector<uint8_t> host;
    std::vector<std::pair<void**, size_t>> at;                         // where each array's device pointer goes, and its offset
    template <typename V>
    void add(const std::vector<V>& v, void** dptr)
    {
        *dptr = nullptr;
        if (v.empty()) return;
        const size_t off = (host.size() + 255) & ~(size_t)255;
        host.resize(off + v.size() * sizeof(V));
        memcpy(host.data() + off, v.data(), v.size() * sizeof(V));
        at.emplace_back(dptr, off);
    }
    size_t tail = 0;                                                   // device-only bytes behind the copied part (reserve)
    std::pair<void**, size_t> tail_at{nullptr, 0};
    void reserve(size_t bytes, void** dptr)                            // space the device fills itself: allocated, not copied
    {
        *dptr = nullptr;
        if (bytes == 0) return;
        tail_at = {dptr, (host.size() + 255) & ~(size_t)255};
        tail = bytes;
    }
    int upload(ohgpu_ctx* ctx, void** slab)
    {
        *slab = nullptr;
        if (host.empty() && tail == 0) return OHGPU_OK;
        const size_t total = tail ? tail_at.second + tail : host.size();
        // (from the context's cache of device blocks: a caller that makes a batch per driver period allocates once)
        hipError_t e = ctx ? ctx_dev_alloc(ctx, slab, total) : hipMalloc(slab, total);
        if (e == hipSuccess && !host.empty()) e = hipMemcpy(*slab, host.data(), host.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && tail) *tail_at.first = (uint8_t*)*slab + tail_at.second;
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE,
                                              "block plan upload: %s", hipGetErrorString(e));
        for (auto& a : at) *a.first = (uint8_t*)*slab + a.second;
        return OHGPU_OK;
    }
};

// a ramp job as the planner carries it: the device's record and the message (the caller's index) whose ramp it applies -- what
// ohgpu_src_batch_set_ramps needs to give the job new endpoints
struct PlanJob : RampJob { uint32_t msg; };

// piece [m_lo, m_hi) of message d (device form dv) for the generic kernel
static DevSrcDesc make_piece(const ohgpu_src_msg_desc& d, uint64_t m_lo, uint64_t m_hi, uint64_t L, uint64_t M)
{
    DevSrcDesc o = src_convert_desc(d, L, M);
    const uint64_t t_first = m_lo * M;
    o.in_rel0 = (int64_t)(t_first / L) - (int64_t)d.src_frame0;
    o.phase0 = (uint32_t)(t_first % L);
    o.n_frames = (uint32_t)(m_hi - m_lo);
    o.ramp_i0 = (uint32_t)(m_lo - d.out_frame0);
    o.ramp_n = d.n_frames;
    o.dst_offset = d.dst_offset + (m_lo - d.out_frame0) * (uint64_t)d.channels * (d.dst_bits / 8);
    return o;
}

// a block: whole phase periods (multiple of L), at least `min_blk` outputs, and a whole number of 64-byte output lines
static uint32_t src_block_outputs_for(uint32_t L, uint32_t fb_dst, uint32_t min_blk)
{
    uint32_t L_blk = L * ((min_blk + L - 1) / L);
    uint32_t k = 1;
    while (k <= 64 && ((uint64_t)L_blk * k * fb_dst) % 64 != 0) k++;
    if (k > 64) return 0;
    return L_blk * k;
}
uint32_t src_block_outputs(uint32_t L, uint32_t fb_dst) { return src_block_outputs_for(L, fb_dst, 128); }

int plan_src_fast(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_src_msg_desc* descs, size_t n, bool ordered, PlanDigest* digest, PlanFusedCheck* fused)
{
    SrcFastPlan& f = b->fast;
    f = SrcFastPlan();
    if (n == 0 || !b->uniform) return OHGPU_OK;
    const ohgpu_src* flt = b->src;
    const uint32_t L = flt->L, M = flt->M, T = flt->T;
    const uint32_t ch = b->channels, sb = b->src_bits / 8, db = b->dst_bits / 8;
    const uint32_t src_le = (b->src_endian == OHGPU_ENDIAN_LITTLE && sb > 1) ? 1 : 0;
    const uint32_t dst_le = (b->dst_endian == OHGPU_ENDIAN_LITTLE) ? 1 : 0;
    // a planar source (OHGPU_FLAG_SRC_PLANAR32) is, per channel, a stream of 4-byte frames: the lean kernel alone reads it
    // (stereo to S24).  Mono, packed 32-bit sources and wide little-endian outputs are lean-only layouts too (round 3):
    // round 1's kernel has no instantiation for them, so such a plan runs on the lean kernel whatever the variant.
    const bool planar = b->src_planar;
    const bool block_ok = !planar && src_block_supported(T, ch, sb, src_le, db, dst_le);
    const bool lean_only = planar ? (T == 32 && ch == 2 && sb != 4 && db == 3) : (!block_ok && src_lean_only_supported(T, ch, sb, src_le, db, dst_le));
    // (a layout only the workgroup matrix kernel has -- big-endian S24 to little-endian S24: planned like a lean-only one, kept only if
    // that kernel takes it, and run on the generic kernel under any variant that asks for another)
    const bool wg_only = !block_ok && !lean_only && !planar && db == 3 &&
                         (((T == 32 || (T == 64 && flt->mf_halfband)) && (ch == 2 || ch == 6 || ch == 8) && sb == 3) || (T == 32 && ch == 2 && sb == 2));
    if (!block_ok && !lean_only && !wg_only) return OHGPU_OK;
    const uint32_t sb_geo = planar ? 3u : sb;                           // (round 1's geometry: only its rows and ring are used)
    const uint32_t sb_lean = planar ? 0u : sb;                          // (LeanGeom: 0 = planar)
    const uint32_t fb_src = planar ? 4u : ch * sb, fb_dst = ch * db;    // (planar: a plane's frame)
    // the ring is drained every four advances: #{j : a <= floor(j*M/L) < a+4} <= ceil(4L/M) outputs arrive in between
    const uint32_t out_per_drain = (4 * L + M - 1) / M;
    uint32_t rows = 0, ring = 0, coef_lds = 0, wave_lds = 0, max_waves = 0;
    if ((block_ok || planar) && !src_block_geometry(L, T, ch, sb_geo, db, out_per_drain, &rows, &ring, &coef_lds, &wave_lds, &max_waves)) return OHGPU_OK;
    // the lean kernel (round 2): same blocks, rows and ring; its rounding bias needs sum|c| < 2^29 in every phase
    uint32_t lean_rows = 0, lean_inb = 0, lean_sf = 8, lean_ring = 0, lean_coef = 0, lean_wave_lds = 0, lean_max_waves = 0;
    const bool lean_hb = flt->halfband && !planar && src_lean_halfband_supported(T, ch, sb, src_le, db, dst_le);
    bool lean = flt->max_sum_abs < ((int64_t)1 << 29) &&
                src_lean_geometry(L, T, lean_hb, ch, sb_lean, db, out_per_drain, &lean_rows, &lean_inb, &lean_sf, &lean_ring, &lean_coef, &lean_wave_lds, &lean_max_waves);
    if (lean && !block_ok && !planar) { rows = lean_rows; ring = lean_ring; }     // (no geometry of round 1's to agree with)
    lean = lean && lean_rows == rows && lean_ring == ring;
    if ((lean_only || wg_only) && !lean) return OHGPU_OK;
    // a block: whole phase periods (multiple of L), at least 128 outputs, and a whole number of 64-byte output lines
    uint32_t min_blk = 128;
#ifdef OHGPU_DIAG
    if (const char* e = getenv("OHGPU_DIAG_MIN_BLOCK")) min_blk = (uint32_t)atoi(e);     // (diagnostic builds: longer blocks per lane)
#endif
    const uint32_t L_blk = src_block_outputs_for(L, fb_dst, min_blk);
    if (L_blk == 0) return OHGPU_OK;
    const uint64_t M_blk64 = (uint64_t)L_blk * M / L;
    if (M_blk64 + T > 32000 || M_blk64 < T) return OHGPU_OK;    // a block is at least one filter length of input
    const uint32_t M_blk = (uint32_t)M_blk64;

    // src_mfma_kernel (round 4) runs the same units for the layouts it serves, from the filter's digit tables -- made for this
    // block length and rows of up to mf_kb_cap blocks -- with its own number of waves per CU.  src_mfma_wg_kernel (the default where
    // it applies; ohgpu_set_kernel_variant(5) keeps the unit-per-wave kernel) takes a unit per WORKGROUP: rows of one block, so that
    // a unit's input and output are each one contiguous run, and no unit whose 32-row input image leaves the arena.
    uint32_t mf_rows = 0, mf_wave_lds = 0, mf_max_waves = 0;
    src_mfma_geometry(&mf_rows, &mf_wave_lds, &mf_max_waves);
    // (mf_layout: 24-bit stereo through the 32-tap tiling -- the layout the unit-per-wave kernel was written for and the workgroup
    // kernel's first; `mfma`: that kernel itself is in this library, which since round 5 it is only in a legacy build)
    const bool mf_layout = lean && !planar && src_mfma_supported(T, ch, sb, db) && flt->d_mf_amat != nullptr && flt->mf_L_blk == L_blk && (L_blk >> 4) <= 16u &&      // (the kernel's bias table: one block's steps)
                           mf_rows == rows && (M_blk + T) * fb_src < (1u << 24);
#ifdef OHGPU_LEGACY_KERNELS
    const bool mfma = mf_layout;
#else
    const bool mfma = false;
#endif
    // (a planar source -- the FLAC decoder's planes -- is the workgroup kernel's too: its split reads the planes; the unit-per-wave
    // kernel has no such form, so with variants 3..5 a planar batch stays on the lean kernel)
    // (six and eight channels too: the same tiles over channel PAIRS, units of 64 / channels rows as the lean kernel's)
    // (and the half-band 2:1 decimator, from its own tables: build_mfma_halfband)
    const bool wg_hb = flt->mf_halfband && T == 64 && !planar;
    const bool wg_tables = lean && flt->d_mf_amat != nullptr && flt->mf_L_blk == L_blk && (T == 32 || wg_hb) && rows == 64u / ch;
    const bool wg_wide = !planar && (ch > 2 || wg_hb || sb == 2) && (M_blk + T) * fb_src < (1u << 24);     // (and 16-bit stereo)
    const bool mfma_wg = wg_tables && (planar || mf_layout || wg_wide) && src_mfma_wg_supported(L_blk, M_blk, ch, sb, db, planar, wg_hb) &&
                         !(ctx && (ctx->variant == 5 || ctx->variant == 3 || ctx->variant == 4 || ctx->variant == 2));
    if (wg_only && !mfma_wg) return OHGPU_OK;
    // The workgroup kernel walks a unit in passes of 16, 5 or 4 rows (WgGeom::kSR) and does not care how many a unit holds: six- and
    // eight-channel units are cut 30 and 32 rows long instead of the lean kernel's 10 and 8 -- a third to a quarter of the units to
    // plan, upload and fetch.  Such a plan is the workgroup kernel's alone (any variant that asks for another gets the generic one).
    const bool wg_long_units = mfma_wg && !planar && ch > 2;
    if (wg_long_units) rows = ch == 6 ? 30u : 32u;

#ifdef OHGPU_PLAN_TIMING
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> tps;
    auto mark = [&](const char* what) { tps.emplace_back(what, std::chrono::steady_clock::now()); };
    mark("start");
#else
    auto mark = [](const char*) {};
#endif
    // order messages by (stream, output position); a stream is identified by where its absolute frame 0 lives
    // (a batch whose messages are in order already -- the usual one -- reads its "order" from a table of 0, 1, 2, .. kept from plan to
    // plan: allocating and filling one per plan was 0.09 of the headline's 1.3 ms.  Up to a million entries; a plan holds the table
    // it got while another thread's larger batch replaces it)
    std::shared_ptr<const std::vector<uint32_t>> identity;
    std::vector<uint32_t> order_own;
    if (ordered && n <= (1u << 20)) {
        static std::mutex mu;
        static std::shared_ptr<const std::vector<uint32_t>> table;
        std::lock_guard<std::mutex> lock(mu);
        if (!table || table->size() < n) {
            auto grown = std::make_shared<std::vector<uint32_t>>(std::max<size_t>(n, 65536));
            std::iota(grown->begin(), grown->end(), 0u);
            table = grown;
        }
        identity = table;
    } else {
        order_own.resize(n);
        std::iota(order_own.begin(), order_own.end(), 0u);
    }
    const uint32_t* const order = identity ? identity->data() : order_own.data();
    auto src_base_of = [&](const ohgpu_src_msg_desc& d) { return (int64_t)d.src_offset - (int64_t)(d.src_frame0 * fb_src); };
    // (planar: two messages of a stream also agree on the distance between its planes; a 4 GiB reach per unit is the lean kernel's)
    if (planar) {
        for (size_t k = 0; k < n; k++)
            if ((uint64_t)(ch - 1) * descs[k].src_plane_stride + (uint64_t)(rows + 1) * (M_blk + T) * 4 + 4096 >= (1ull << 32)) return OHGPU_OK;
    }
    auto dst_base_of = [&](const ohgpu_src_msg_desc& d) { return (int64_t)d.dst_offset - (int64_t)(d.out_frame0 * fb_dst); };
    auto before = [&](uint32_t x, uint32_t y) {
        const int64_t sx = src_base_of(descs[x]), sy = src_base_of(descs[y]);
        if (sx != sy) return sx < sy;
        if (descs[x].src_plane_stride != descs[y].src_plane_stride) return descs[x].src_plane_stride < descs[y].src_plane_stride;
        const int64_t dx = dst_base_of(descs[x]), dy = dst_base_of(descs[y]);
        if (dx != dy) return dx < dy;
        return descs[x].out_frame0 < descs[y].out_frame0;
    };
    // (a caller that lists its streams one after the other, each in time order -- the usual case -- is in order already, and the
    // validation pass has seen that: no pass of this planner's, no sort of half a million messages)
    if (!ordered) std::sort(order_own.begin(), order_own.end(), before);

    mark("order");
    struct SegRun { uint32_t seg; uint64_t blk_lo, blk_hi; uint32_t work_begin; };   // a segment's whole blocks and where its units start in `work`
    // what a stretch of messages contributes (indices local to the stretch until the stretches are put together)
    struct Stretch {
        std::vector<SrcSeg> segs;
        std::vector<SegRun> seg_runs;
        std::vector<uint32_t> seg_plane_stride;         // planar batches: bytes between a segment's planes
        std::vector<SrcWork> work;
        std::vector<DevSrcDesc> rem;
        std::vector<uint32_t> rem_msg;                  // ... and the message (the caller's index) each piece is a part of
        bool stream_start = false;                      // a message whose filter window reaches in front of its stream's first frame
        uint64_t fast_frames = 0;
        // the workgroup kernel's plan (every unit one of `work`'s, rows of one block): its units, ramp jobs and planes are made where
        // the work units are, in the same pass over the messages -- indices local to the stretch, as everything here
        std::vector<LeanUnit> units;
        std::vector<PlanJob> jobs;
        size_t planes = 0;
        bool ok = true;
        // the fused check (PlanFusedCheck): what the stretch's messages were found to be, and the next one not yet looked at
        SrcRangeResult chk;
        size_t checked_upto = 0, check_end = 0;
        // the message last visited (`visit` below): its position and its stream's bases; and, for the order test, its predecessor's
        size_t vis_k = (size_t)-1;
        int64_t vis_sb = 0, vis_db = 0;
    };
    // lean kernel: one plane of multipliers per ramped unit -- rows * L_blk entries (uint16, 0xffff = no ramp on that frame)
    // (the kernel loads eight entries at a time; a row is a whole number of loads when L_blk is a multiple of 8, else the slack covers the last one)
    // (the multipliers themselves are computed on the device, csrc/ramp_plane_kernel.hip: the planner only says which message
    // covers which stretch of which plane)
    std::vector<PlanJob> ramp_jobs;
    size_t plane_entries = 0;

    // the lean kernel's staging moves, per stage q, the aligned 16-byte pieces that hold each row's frames: does every one of
    // them lie inside the arena?  (Only a unit at an end of the arena can fail.)  A row is `kb` consecutive blocks.
    auto unit_leaves_arena = [&](int64_t sbase, uint64_t plane_stride, uint64_t first_block, uint32_t n_rows, uint32_t kb) {
        const int64_t m_row = (int64_t)M_blk * kb;
        const int64_t total = m_row + T, n_stages = (total + lean_sf - 1) / lean_sf;
        for (uint32_t pc = 0; pc < (planar ? ch : 1u); pc++) {              // (planar: every channel's plane is staged on its own)
            const int64_t g_first = sbase + (int64_t)(pc * plane_stride) + ((int64_t)(first_block * M_blk) - (int64_t)T) * fb_src;
            const int64_t g_last = g_first + (int64_t)(n_rows - 1) * m_row * fb_src;
            const int64_t lo = g_first - (g_first & 15);
            const int64_t hi = g_last - (g_last & 15) + 16 * (((g_last & 15) + lean_sf * fb_src + 15) >> 4) + (n_stages - 1) * lean_sf * fb_src;
            if (g_first < 0 || lo < 0 || (uint64_t)hi > b->src_arena_bytes) return true;
        }
        return false;
    };

    // Message k continues message k - 1 -- same stream, the next output frame -- or starts a segment: a test on the pair alone, so
    // the messages can be cut into stretches of whole segments and the stretches planned side by side; put together in order they
    // are the arrays one thread walking all the messages makes.
    auto continues = [&](size_t k) {
        const ohgpu_src_msg_desc& p = descs[order[k - 1]];
        const ohgpu_src_msg_desc& d = descs[order[k]];
        return p.n_frames != 0 && d.n_frames != 0 && src_base_of(d) == src_base_of(p) && dst_base_of(d) == dst_base_of(p) &&
               d.out_frame0 == p.out_frame0 + p.n_frames && d.src_plane_stride == p.src_plane_stride;
    };
    // One lean / matrix-kernel unit: `n_rows` rows of `kb` consecutive blocks from block `bk` of the stream whose absolute frame 0
    // lives at (sbase, dbase); `flags` = what is known of it already (kWorkRamped | kWorkChecked), `mi` .. `msg_end` the positions in
    // `order` of the messages from the one that holds its first output frame on.  A ramped unit gets a plane of multipliers -- n_rows *
    // L_blk entries (uint16, 0xffff = no ramp on that frame), in whole 16-byte pieces; the kernel addresses a plane as planes + plane *
    // plane_stride with a stride of 16 -- and one job per ramped message that reaches into it.
    auto emit_unit = [&](std::vector<LeanUnit>& units_out, std::vector<PlanJob>& jobs_out, size_t& planes_out, int64_t sbase, int64_t dbase,
                         uint32_t plane_stride_bytes, uint64_t bk, uint32_t n_rows, uint32_t kb, uint32_t flags, uint32_t mi, uint32_t msg_end) -> bool {
        LeanUnit u;
        u.src_row0 = sbase + ((int64_t)(bk * M_blk) - (int64_t)T) * fb_src;
        u.dst_row0 = dbase + (int64_t)(bk * L_blk) * fb_dst;
        u.n_blocks = n_rows;
        u.flags = (kb << 8) | (bk == 0 ? (uint32_t)kWorkFirst : 0u) | (flags & (kWorkRamped | kWorkChecked));
        u.plane = 0;
        u.src_plane_stride = planar ? plane_stride_bytes : 0u;
        // the workgroup kernel reads 32 rows' worth of input per unit whatever the unit holds and checks nothing: the (at most two)
        // units of a batch for which that leaves the arena fetch their pieces through the checked load
        if (mfma_wg && !src_mfma_wg_unit_inside(u.src_row0, M_blk * fb_src, b->src_arena_bytes, ch, sb, rows, planar, (uint64_t)(ch - 1) * u.src_plane_stride, wg_hb)) u.flags |= kWorkEdge;
        if (u.flags & kWorkRamped) {
            const uint64_t u_lo = bk * L_blk, u_hi = (bk + (uint64_t)n_rows * kb) * L_blk;
            const size_t per = (((size_t)n_rows * kb * L_blk + 8) + 7) & ~(size_t)7;
            if (planes_out / 8 + per / 8 > 0xffffffffull) return false;
            u.plane = (uint32_t)(planes_out / 8);
            for (uint32_t m = mi; m < msg_end && descs[order[m]].out_frame0 < u_hi; m++) {
                const ohgpu_src_msg_desc& d = descs[order[m]];
                if (!(d.flags & OHGPU_FLAG_RAMP)) continue;
                const uint64_t lo = std::max<uint64_t>(d.out_frame0, u_lo), hi = std::min<uint64_t>(d.out_frame0 + d.n_frames, u_hi);
                if (hi <= lo) continue;
                PlanJob j;
                memset(&j, 0, sizeof(j));
                j.msg = order[m];
                j.plane_entry = planes_out + (lo - u_lo);
                j.i0 = (uint32_t)(lo - d.out_frame0); j.count = (uint32_t)(hi - lo); j.n = d.n_frames;
                uint32_t sh = 0;                        // RampApplicator divides by n - 1 per frame (Msg.cpp:835): exact multiplier instead
                magic_u31(d.n_frames > 1 ? d.n_frames - 1 : 1, &j.m_n1, &sh);
                j.s_n1 = (uint8_t)sh; j.ramp_start = d.ramp_start; j.ramp_end = d.ramp_end;
                jobs_out.push_back(j);
            }
            planes_out += per;
        }
        units_out.push_back(u);
        return true;
    };
    // (the workgroup kernel's units are `work`'s one for one -- rows of one block, kb_max = 1 below -- and round 1's arrays are not
    // made for such a plan: its units come out of the pass that finds them)
    const bool direct_units = mfma_wg;
    // (fused check: a message is validated the first time the pass looks at it -- in message order, each once; a stretch stops at its
    // first bad descriptor, or at the first that is not of the batch's layout or out of order)
    // (... a run of them at a time: one call and its set-up per 128 messages, not per message.  A bad descriptor further on in the run
    // stops the stretch before the pass reaches it -- the plan is thrown away either way, and the first bad one is still the first)
#ifdef OHGPU_PLAN_TIMING
    static thread_local uint64_t tsc_check, tsc_grow, tsc_units, tsc_rem;
    tsc_check = tsc_grow = tsc_units = tsc_rem = 0;
#define PLAN_TSC(acc, stmt) { const uint64_t t0_ = __builtin_ia32_rdtsc(); stmt; acc += __builtin_ia32_rdtsc() - t0_; }
#else
#define PLAN_TSC(acc, stmt) { stmt; }
#endif
    auto looked_at = [&](Stretch& o, size_t k) -> bool {
        if (!fused || k < o.checked_upto) return true;
        const size_t upto = std::min(k + 128, o.check_end);
        PLAN_TSC(tsc_check, src_check_range(fused->src, descs, k, upto, b->src_arena_bytes, b->dst_arena_bytes, nullptr, &o.chk));
        o.checked_upto = upto;
        return o.chk.err == OHGPU_OK && o.chk.uniform && o.chk.ordered;
    };
    // The pass's look at message k: its stream's bases, and -- fused check -- whether the pass may go on.  The usual message (packed
    // source, descs[0]'s layout, nothing wrong) is checked in line (SrcQuickCheck) and tested against its predecessor for the planner's
    // order with the bases both need anyway; any other goes to src_check_range, message by message (planar batches: a run at a time).
    const bool quick_ok = fused && !planar && n > 0;
    const SrcQuickCheck quick(fused ? fused->src->L : L, fused ? fused->src->M : M, fused ? fused->src->T : T, descs[0], b->src_arena_bytes, b->dst_arena_bytes);
    auto visit = [&](Stretch& o, size_t k, int64_t* sb, int64_t* db) __attribute__((always_inline)) -> bool {
        if (k == o.vis_k) { *sb = o.vis_sb; *db = o.vis_db; return true; }
        const ohgpu_src_msg_desc& d = descs[order[k]];
        if (!quick_ok) {
            if (!looked_at(o, k)) return false;
            *sb = src_base_of(d); *db = dst_base_of(d);
        } else {
            // (fused: order[] is the identity -- the caller's order is the claim being checked)
            bool slow = !quick.pass(d, o.chk, sb, db);
            if (!slow && k > 0) {
                // the planner's order against the predecessor (src_msg_before): the last message visited, or -- a stretch's first -- the
                // one in front of the stretch
                int64_t psb, pdb;
                const ohgpu_src_msg_desc& pd = descs[k - 1];
                if (k - 1 == o.vis_k) { psb = o.vis_sb; pdb = o.vis_db; }
                else { psb = src_base_of(pd); pdb = dst_base_of(pd); }
                const bool before = *sb != psb ? *sb < psb : (*db != pdb ? *db < pdb : d.out_frame0 < pd.out_frame0);
                if (before) o.chk.ordered = false;
            }
            if (slow) {
                PLAN_TSC(tsc_check, src_check_range(fused->src, descs, k, k + 1, b->src_arena_bytes, b->dst_arena_bytes, nullptr, &o.chk));
                *sb = src_base_of(d); *db = dst_base_of(d);
            }
            if (o.chk.err != OHGPU_OK || !o.chk.uniform || !o.chk.ordered) return false;
        }
        o.vis_k = k; o.vis_sb = *sb; o.vis_db = *db;
        return true;
    };
    auto plan_stretch = [&](size_t i_begin, size_t i_end, Stretch& o) {
            size_t i = i_begin;
            o.checked_upto = i_begin;
            o.check_end = i_end;
            while (i < i_end) {
            // grow a run of messages that tile a contiguous output range of one stream
            size_t e = i + 1;
            const ohgpu_src_msg_desc& d0 = descs[order[i]];
            int64_t sbase, dbase;
            if (!visit(o, i, &sbase, &dbase)) return;
            if (d0.n_frames != 0 && (d0.out_frame0 * M) / L < T - 1u) o.stream_start = true;     // (a run's first message reaches furthest back)
            uint64_t next_out = d0.out_frame0 + d0.n_frames;
            bool zero_len = d0.n_frames == 0;
#ifdef OHGPU_PLAN_TIMING
            const uint64_t tg0 = __builtin_ia32_rdtsc();
#endif
            while (!zero_len && e < i_end) {
                int64_t sb_e, db_e;
                __builtin_prefetch(&descs[order[e + 32 < i_end ? e + 32 : e]]);
                if (!visit(o, e, &sb_e, &db_e)) return;
                const ohgpu_src_msg_desc& d = descs[order[e]];
                if (d.n_frames == 0 || sb_e != sbase || db_e != dbase || d.out_frame0 != next_out ||
                    d.src_plane_stride != d0.src_plane_stride) break;
                next_out += d.n_frames;
                e++;
            }
#ifdef OHGPU_PLAN_TIMING
            const uint64_t tg1 = __builtin_ia32_rdtsc();
            tsc_grow += tg1 - tg0;
#endif
            const uint64_t m_begin = d0.out_frame0, m_end = next_out;
            uint64_t blk_lo = (m_begin + L_blk - 1) / L_blk, blk_hi = m_end / L_blk;
            // (a stream whose output does not start on a 64-byte boundary is written with unaligned 16-byte stores: they run at
            // the aligned rate -- tools/micro/unaligned_store.hip -- only the lines are then no longer whole)
            bool fast_ok = !zero_len && blk_hi > blk_lo;
            if (fast_ok) {
                // every whole block's history must be present in the windows the caller declared (they were validated
                // per message; the block reads nothing a message of the block does not itself need)
                // (a message is known by its position in `order`: the segment's are [i, e).  Their ramp parameters are read from the
                // descriptors where a unit needs them; round 1's kernel gets them as an array, made when that kernel is planned for)
                SrcSeg sg;
                sg.src_base = sbase; sg.dst_base = dbase; sg.msg_begin = (uint32_t)i; sg.msg_end = (uint32_t)e;
                const uint32_t seg_index = (uint32_t)o.segs.size();
                o.segs.push_back(sg);
                o.seg_plane_stride.push_back((uint32_t)d0.src_plane_stride);
                o.seg_runs.push_back(SegRun{seg_index, blk_lo, blk_hi, (uint32_t)o.work.size()});
                uint32_t mi = (uint32_t)i;                    // message that holds the unit's first output frame
                while (mi + 1 < sg.msg_end && descs[order[mi + 1]].out_frame0 <= blk_lo * L_blk) mi++;
                const uint32_t* const ord = order;
                for (uint64_t bk = blk_lo; bk < blk_hi; bk += rows) {
                    SrcWork w;
                    w.first_block = bk; w.seg = seg_index; w.n_blocks = (uint32_t)std::min<uint64_t>(rows, blk_hi - bk);
                    w.msg_first = mi;
                    // cost class, for the order below: a wave that meets a ramped message goes through the per-output ramp path
                    // for all of its lanes, which makes such a unit two to three times as long as a plain one
                    // (ONE walk over the unit's messages finds that and the next unit's first message -- the last that starts at or
                    // before the unit's end: within a segment every message is longer than nothing and they come in output order)
                    const uint64_t u_lo = bk * L_blk, u_hi = (bk + w.n_blocks) * L_blk;
                    bool ramped = false;
                    uint32_t m = mi;
                    for (; m < sg.msg_end && descs[ord[m]].out_frame0 < u_hi; m++) {
                        const ohgpu_src_msg_desc& dm = descs[ord[m]];
                        ramped |= (dm.flags & OHGPU_FLAG_RAMP) && dm.out_frame0 + dm.n_frames > u_lo;
                    }
                    const uint32_t mi_next = m < sg.msg_end && descs[ord[m]].out_frame0 == u_hi ? m : m - 1u;
                    w.flags = ramped ? kWorkRamped : 0u;
                    w.plane = 0; w.pad = 0;
                    if (unit_leaves_arena(sbase, d0.src_plane_stride, bk, w.n_blocks, 1)) w.flags |= kWorkChecked;
                    if (direct_units) o.ok = o.ok && emit_unit(o.units, o.jobs, o.planes, sbase, dbase, (uint32_t)d0.src_plane_stride, bk, w.n_blocks, 1, w.flags, mi, sg.msg_end);
                    else o.work.push_back(w);
                    mi = mi_next;
                }
                o.fast_frames += (blk_hi - blk_lo) * L_blk;
            } else {
                blk_lo = blk_hi = 0;   // everything goes to the generic kernel
            }
            const uint64_t fast_lo = fast_ok ? blk_lo * L_blk : m_end, fast_hi = fast_ok ? blk_hi * L_blk : m_end;
#ifdef OHGPU_PLAN_TIMING
            const uint64_t tg2 = __builtin_ia32_rdtsc();
            tsc_units += tg2 - tg1;
#endif
            for (size_t k = i; k < e; k++) {
                const ohgpu_src_msg_desc& d = descs[order[k]];
                if (d.n_frames == 0) continue;
                const uint64_t lo = d.out_frame0, hi = d.out_frame0 + d.n_frames;
                if (!fast_ok) { o.rem.push_back(make_piece(d, lo, hi, L, M)); o.rem_msg.push_back(order[k]); continue; }
                if (lo < fast_lo) { o.rem.push_back(make_piece(d, lo, std::min(hi, fast_lo), L, M)); o.rem_msg.push_back(order[k]); }
                if (hi > fast_hi) { o.rem.push_back(make_piece(d, std::max(lo, fast_hi), hi, L, M)); o.rem_msg.push_back(order[k]); }
            }
#ifdef OHGPU_PLAN_TIMING
            tsc_rem += __builtin_ia32_rdtsc() - tg2;
#endif
            i = e;
        }
    };
    // (four stretches a thread: the threads claim them as they go, so one that wakes late or shares its core takes fewer)
    const unsigned stretch_threads = plan_threads(n, 32768);
    const unsigned n_stretch = stretch_threads <= 1 ? 1u : (unsigned)std::min<size_t>(4u * stretch_threads, std::max<size_t>(n / 8192, stretch_threads));
    std::vector<Stretch> parts(n_stretch);
    {
        std::vector<size_t> cut(n_stretch + 1, n);
        cut[0] = 0;
        for (unsigned t = 1; t < n_stretch; t++) {
            size_t k = n * t / n_stretch;
            while (k < n && k > 0 && continues(k)) k++;                  // (forward to the next segment's first message)
            cut[t] = k;
        }
        for (unsigned t = 1; t <= n_stretch; t++) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
        parallel_ranges(n_stretch, n_stretch, stretch_threads, [&](unsigned, size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; t++) if (cut[t] < cut[t + 1]) plan_stretch(cut[t], cut[t + 1], parts[t]);
        });
    }
    if (fused) {
        // (stretches are in message order and each stopped at its first bad descriptor: the first stretch with one holds the batch's first)
        for (const Stretch& o : parts) {
            if (o.chk.err != OHGPU_OK) { fused->total.err = o.chk.err; memcpy(fused->total.msg, o.chk.msg, sizeof(o.chk.msg)); break; }
            if (!o.chk.uniform || !o.chk.ordered) { fused->retry = true; break; }
            fused->total.in_frames += o.chk.in_frames; fused->total.out_frames += o.chk.out_frames;
            fused->total.src_bytes_touched += o.chk.src_bytes_touched; fused->total.dst_bytes_written += o.chk.dst_bytes_written;
            if (o.chk.max_frames > fused->total.max_frames) fused->total.max_frames = o.chk.max_frames;
        }
        fused->checked = true;
        if (fused->total.err != OHGPU_OK || fused->retry) return OHGPU_OK;      // (the caller reports the error, or takes the two-pass route)
    }
    std::vector<SrcSeg> segs;
    std::vector<SegRun> seg_runs;
    std::vector<uint32_t> seg_plane_stride;
    std::vector<SrcWork> work;
    std::vector<DevSrcDesc> rem;
    std::vector<uint32_t> rem_msg;
    bool stream_start = false;
    uint64_t fast_frames = 0;
    std::vector<LeanUnit> lean_units;
    {
        size_t n_segs = 0, n_work0 = 0, n_rem = 0, n_units0 = 0, n_jobs0 = 0;
        for (const Stretch& o : parts) { n_segs += o.segs.size(); n_work0 += o.work.size(); n_rem += o.rem.size(); n_units0 += o.units.size(); n_jobs0 += o.jobs.size(); }
        segs.reserve(n_segs); seg_runs.reserve(n_segs); seg_plane_stride.reserve(n_segs); work.reserve(n_work0); rem.reserve(n_rem);
        lean_units.reserve(n_units0); ramp_jobs.reserve(n_jobs0);
        for (size_t t = 0; t < parts.size(); t++) {
            Stretch& o = parts[t];
            const uint32_t seg_base = (uint32_t)segs.size(), work_base = (uint32_t)work.size();
            segs.insert(segs.end(), o.segs.begin(), o.segs.end());
            for (SegRun r : o.seg_runs) { r.seg += seg_base; r.work_begin += work_base; seg_runs.push_back(r); }
            seg_plane_stride.insert(seg_plane_stride.end(), o.seg_plane_stride.begin(), o.seg_plane_stride.end());
            for (SrcWork w : o.work) { w.seg += seg_base; work.push_back(w); }
            rem.insert(rem.end(), o.rem.begin(), o.rem.end());
            rem_msg.insert(rem_msg.end(), o.rem_msg.begin(), o.rem_msg.end());
            stream_start = stream_start || o.stream_start;
            fast_frames += o.fast_frames;
            // (the workgroup kernel's units: a stretch's planes and jobs count from zero and go behind its predecessors')
            if (!o.ok || (plane_entries + o.planes) / 8 > 0xffffffffull) return OHGPU_OK;
            const uint32_t plane_base = (uint32_t)(plane_entries / 8);
            for (LeanUnit& u : o.units) { if (u.flags & kWorkRamped) u.plane += plane_base; lean_units.push_back(u); }
            for (PlanJob& j : o.jobs) { j.plane_entry += plane_entries; ramp_jobs.push_back(j); }
            plane_entries += o.planes;
        }
        parts.clear();
    }
    if (direct_units ? lean_units.empty() : work.empty()) return OHGPU_OK;
    mark("segments");
#ifdef OHGPU_PLAN_TIMING
    fprintf(stderr, "[plan timing]   (this thread's cycles: check %.2f M, grow less check %.2f M, units %.2f M, remainder %.2f M)\n",
            tsc_check * 1e-6, (tsc_grow - tsc_check) * 1e-6, tsc_units * 1e-6, tsc_rem * 1e-6);
#endif
    // ---- the lean kernel's units.  A unit is `rows` rows; a row is `kb` CONSECUTIVE blocks of its stream.  With kb = 1 (round
    // 2) every block pays a filter length of warm-up advances and re-reads that much history (32 frames per 147), and every
    // 160 outputs a unit set-up; a row of kb blocks pays them once.  But long units make the end of the launch coarse -- round
    // 2 measured uniformly longer blocks as a loss for exactly that reason.  So: ONE LONG UNIT PER WAVE, as long as the plain
    // work allows (claimed first: every wave starts on one), everything else -- what does not divide, and the ramped units,
    // which stay one block long because a long unit would run the ramp path for all its outputs -- as one-block units for
    // the waves to level out on.
    if (lean) {
        const uint32_t waves = (ctx && ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u) * (mfma ? mf_max_waves : lean_max_waves);
        // Same-box A/Bs on the headline workload, alternating passes, +-0.2 % within a box.  Eleven waves per CU
        // (tools/exp_units3.sh): one block per row everywhere 0.4955 ms; one long unit of 8 blocks per wave 0.4832; 6 blocks and
        // two rounds of short units kept 0.4995; two rounds of 4-block units 0.5025; three rounds of 2-block units 0.4933.
        // Twelve waves (tools/exp_units4.sh): one block per row 0.496; 7-block units for eleven of twelve waves 0.4475; for
        // EVERY wave (3072 long units for 3072 waves, 0.83 rounds of short ones left) 0.4375; 6-block units 0.50-0.51.  What a
        // long row saves by itself is small (set-up and warm-up are 7 % of a one-block unit; the history it does not re-read is
        // 18 % of the reads); what decides is how the schedule's last units fall, and "every wave exactly one long unit, as
        // long as possible" is the rule that was best at both occupancies.
        uint32_t kb_max = 8, long_rounds = 0;
        double tail_rounds = 0.0;                            // (diagnostic: plain work held back from the long units, in rounds of short units)
#ifdef OHGPU_DIAG
        if (const char* e = getenv("OHGPU_DIAG_TAIL_ROUNDS")) tail_rounds = atof(e);       // (diagnostic builds: the long/short split)
        if (const char* e = getenv("OHGPU_DIAG_KB_MAX")) kb_max = (uint32_t)atoi(e);
        if (const char* e = getenv("OHGPU_DIAG_LONG_ROUNDS")) long_rounds = (uint32_t)atoi(e);
#endif
        if (mfma && kb_max > flt->mf_kb_cap) kb_max = flt->mf_kb_cap;      // (the step table's length)
        if (mfma_wg) kb_max = 1;                                           // (a unit = consecutive blocks)
        // the plain, full one-block units (the only ones that merge), in runs between ramped or partly filled ones
        uint64_t plain_total = 0;
        for (const SrcWork& w : work) plain_total += (!(w.flags & kWorkRamped) && w.n_blocks == rows) ? 1u : 0u;
        const double plain_avail = (double)plain_total - tail_rounds * waves;
        uint32_t kb_long = 1;
        uint64_t long_target = 0;                            // long units to cut, over all segments
        if (ctx && ctx->variant == 3 && !mfma_wg) {        // ohgpu_set_kernel_variant(3): the long rows forced, for tests with small batches
            kb_long = 3;
            long_target = ~(uint64_t)0;
        } else if (plain_avail >= 2.0 * waves && kb_max >= 2) {
            // rounds of long units: one, unless even 8-block units would leave more than that for a second helping
            const uint32_t n_rounds = long_rounds ? long_rounds : (uint32_t)std::max(1.0, std::ceil(plain_avail / ((double)waves * kb_max) - 0.25));
            kb_long = (uint32_t)(plain_avail / ((double)waves * n_rounds));
            if (kb_long > kb_max) kb_long = kb_max;
            if (kb_long < 2) kb_long = 1;
            long_target = kb_long > 1 ? (uint64_t)waves * n_rounds : 0;
        }
        uint64_t long_cut = 0;
        if (!direct_units) lean_units.reserve(work.size());
        // (a unit goes to `units_out`, its ramp jobs to `jobs_out`, its plane behind `planes_out` entries: the batch's own arrays, or a
        // thread's share of them that is put behind the others' afterwards)
        auto emit_to = [&](std::vector<LeanUnit>& units_out, std::vector<PlanJob>& jobs_out, size_t& planes_out,
                           const SegRun& r, uint64_t bk, uint32_t n_rows, uint32_t kb, const SrcWork* w1) -> bool {
            uint32_t flags = 0;
            if (w1) flags = w1->flags & (kWorkRamped | kWorkChecked);
            else if (unit_leaves_arena(segs[r.seg].src_base, planar ? seg_plane_stride[r.seg] : 0u, bk, n_rows, kb)) flags = kWorkChecked;
            return emit_unit(units_out, jobs_out, planes_out, segs[r.seg].src_base, segs[r.seg].dst_base, seg_plane_stride[r.seg], bk, n_rows, kb, flags,
                             w1 ? w1->msg_first : segs[r.seg].msg_begin, segs[r.seg].msg_end);
        };
        auto emit = [&](const SegRun& r, uint64_t bk, uint32_t n_rows, uint32_t kb, const SrcWork* w1) -> bool {
            return emit_to(lean_units, ramp_jobs, plane_entries, r, bk, n_rows, kb, w1);
        };
        if (direct_units) {
            // (made with the segments, above)
        } else if (kb_long == 1 && seg_runs.size() >= 64) {
            // every unit is one of `work`'s: the segments' runs in ranges, side by side; a range's planes and jobs count from zero and
            // are moved behind its predecessors' when the ranges are put together, in order -- the arrays one thread makes
            struct Share { std::vector<LeanUnit> units; std::vector<PlanJob> jobs; size_t planes = 0; bool ok = true; };
            const unsigned n_thr = plan_threads(work.size(), 2048);
            std::vector<Share> shares(n_thr);
            parallel_ranges(seg_runs.size(), n_thr, [&](unsigned t, size_t lo, size_t hi) {
                Share& sh = shares[t];
                for (size_t ri = lo; ri < hi && sh.ok; ri++) {
                    const SegRun& r = seg_runs[ri];
                    const uint32_t n_units = (uint32_t)((r.blk_hi - r.blk_lo + rows - 1) / rows);
                    for (uint32_t k = 0; k < n_units && sh.ok; k++) {
                        const SrcWork& w1 = work[r.work_begin + k];
                        sh.ok = emit_to(sh.units, sh.jobs, sh.planes, r, w1.first_block, w1.n_blocks, 1, &w1);
                    }
                }
            });
            for (Share& sh : shares) {
                if (!sh.ok || (plane_entries + sh.planes) / 8 > 0xffffffffull) return OHGPU_OK;
                const uint32_t plane_base = (uint32_t)(plane_entries / 8);
                for (LeanUnit& u : sh.units) { if (u.flags & kWorkRamped) u.plane += plane_base; lean_units.push_back(u); }
                for (PlanJob& j : sh.jobs) { j.plane_entry += plane_entries; ramp_jobs.push_back(j); }
                plane_entries += sh.planes;
            }
        } else for (const SegRun& r : seg_runs) {
            const uint32_t n_units = (uint32_t)((r.blk_hi - r.blk_lo + rows - 1) / rows);
            uint32_t k = 0;
            while (k < n_units) {
                const SrcWork& w1 = work[r.work_begin + k];
                // a run of plain, full one-block units: long units from its front, the rest (what does not divide) stays short
                uint32_t run = 0;
                while (kb_long > 1 && k + run < n_units && !(work[r.work_begin + k + run].flags & kWorkRamped) &&
                       work[r.work_begin + k + run].n_blocks == rows) run++;
                // (as many long units as the run holds, until every wave has its one)
                const uint32_t n_long = kb_long > 1 ? (uint32_t)std::min<uint64_t>(run / kb_long, long_target - long_cut) : 0u;
                if (n_long > 0) {
                    long_cut += n_long;
                    for (uint32_t q = 0; q < n_long; q++)
                        if (!emit(r, w1.first_block + (uint64_t)q * kb_long * rows, rows, kb_long, nullptr)) return OHGPU_OK;
                    k += n_long * kb_long;
                    for (uint32_t q = n_long * kb_long; q < run; q++, k++)
                        if (!emit(r, work[r.work_begin + k].first_block, rows, 1, &work[r.work_begin + k])) return OHGPU_OK;
                    continue;
                }
                // (nothing long comes out of this run -- the target is met, or the run is shorter than a long unit: all of it as
                // one-block units in one pass, not one rescan of the run per unit)
                for (uint32_t q = 0; q < std::max(run, 1u); q++, k++)
                    if (!emit(r, work[r.work_begin + k].first_block, work[r.work_begin + k].n_blocks, 1, &work[r.work_begin + k])) return OHGPU_OK;
            }
        }
        if (plane_entries) plane_entries += (size_t)rows * L_blk + 8;                             // (lanes without a block read their row's place too)
        // Longest first: the waves claim units in this order, and the kernel ends when the last unit does.
        {   // (a stable sort by descending cost; the costs are a handful of values, so: one bucket per value, in order)
            auto cost = [](const LeanUnit& u) { return ((u.flags & kWorkRamped) ? 6u : 5u) * ((u.flags >> 8) & 0xffu) * u.n_blocks; };
            std::vector<uint32_t> costs;
            for (const LeanUnit& u : lean_units) { const uint32_t c = cost(u); if (std::find(costs.begin(), costs.end(), c) == costs.end()) costs.push_back(c); }
            if (costs.size() <= 64) {
                std::sort(costs.begin(), costs.end(), std::greater<uint32_t>());
                std::vector<LeanUnit> sorted;
                sorted.reserve(lean_units.size());
                for (uint32_t c : costs) for (const LeanUnit& u : lean_units) if (cost(u) == c) sorted.push_back(u);
                lean_units.swap(sorted);
            } else {
                std::stable_sort(lean_units.begin(), lean_units.end(), [&](const LeanUnit& x, const LeanUnit& y) { return cost(x) > cost(y); });
            }
        }
        // (the edge units in front: their checked loads are slow, and the launch should not end on them)
        if (mfma_wg) std::stable_partition(lean_units.begin(), lean_units.end(), [](const LeanUnit& u) { return (u.flags & kWorkEdge) != 0; });
    }
    // (segments, messages and one-block work units are round 1's kernel's: they go to the device only for a batch planned while
    // ohgpu_set_kernel_variant(2) is in force, or one the lean kernel cannot run -- 12 MB of the headline's plan, and most of the time its upload took)
    // (a filter the lean kernel's rounding does not hold -- sum|c| >= 2^29 -- runs on round 1's whatever the variant, where this
    // library has that kernel for the layout: five stereo layouts in the shipped library, the whole list in a legacy build, which
    // also takes it under variant 2)
    const bool round1_built = block_ok && src_block_built(T, ch, sb, src_le, db, dst_le);
#ifdef OHGPU_LEGACY_KERNELS
    const bool round1 = round1_built && ((ctx && ctx->variant == 2) || !lean);
#else
    const bool round1 = round1_built && !lean;
#endif
    if (!lean && !round1) return OHGPU_OK;                                      // (neither block kernel can take it: the generic kernel's batch)
    // (round 1's kernel, variant 2, keeps one-block units; ramped first, partly filled units last)
    if (round1)
        std::stable_sort(work.begin(), work.end(), [](const SrcWork& x, const SrcWork& y) {
            return ((x.flags & kWorkRamped) ? 3u : 1u) * x.n_blocks > ((y.flags & kWorkRamped) ? 3u : 1u) * y.n_blocks;
        });

    mark("units");
    const std::vector<RampJob> dev_jobs(ramp_jobs.begin(), ramp_jobs.end());        // (the device's records)
    if (digest) {
        // (ohgpu_src_plan_digest: what the plan consists of, hashed; nothing goes to a device)
        auto fnv = [](uint64_t h, const void* p, size_t bytes) {
            const uint8_t* q = (const uint8_t*)p;
            for (size_t k = 0; k < bytes; k++) h = (h ^ q[k]) * 1099511628211ull;
            return h;
        };
        uint64_t h = 1469598103934665603ull;
        h = fnv(h, lean_units.data(), lean_units.size() * sizeof(LeanUnit));
        if (round1) {                                       // (round 1's arrays are part of a plan only where they go to the device)
            h = fnv(h, work.data(), work.size() * sizeof(SrcWork));
            h = fnv(h, segs.data(), segs.size() * sizeof(SrcSeg));
        }
        h = fnv(h, rem.data(), rem.size() * sizeof(DevSrcDesc));
        h = fnv(h, dev_jobs.data(), dev_jobs.size() * sizeof(RampJob));
        h = fnv(h, &plane_entries, sizeof(plane_entries));
        digest->hash = h;
        digest->units = lean ? lean_units.size() : work.size();
        digest->pieces = rem.size();
        digest->ramp_jobs = ramp_jobs.size();
        digest->kernel = mfma_wg ? 3 : (mfma ? 2 : (lean ? 1 : 0));
#ifdef OHGPU_PLAN_TIMING
        for (size_t k = 1; k < tps.size(); k++)
            fprintf(stderr, "[plan timing]   %s %.2f ms\n", tps[k].first, std::chrono::duration<double, std::milli>(tps[k].second - tps[k - 1].second).count());
#endif
        return OHGPU_OK;
    }
    Slab slab;
    if (round1) {
        // the messages' ramp parameters as an array, by position in `order` (SrcSeg::msg_begin .. msg_end, SrcWork::msg_first)
        std::vector<SegMsg> msgs(n);
        parallel_ranges(n, plan_threads(n, 32768), [&](unsigned, size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; k++) {
                const ohgpu_src_msg_desc& d = descs[order[k]];
                SegMsg& sm = msgs[k];
                memset(&sm, 0, sizeof(sm));
                sm.out0 = d.out_frame0; sm.n = d.n_frames; sm.ramp_start = d.ramp_start; sm.ramp_end = d.ramp_end; sm.flags = d.flags;
                if (d.flags & OHGPU_FLAG_RAMP) {            // RampApplicator divides by n - 1 per frame (Msg.cpp:835): exact multiplier instead
                    uint32_t sh = 0;
                    magic_u31(d.n_frames > 1 ? d.n_frames - 1 : 1, &sm.m_n1, &sh);
                    sm.s_n1 = (uint8_t)sh;
                }
            }
        });
        slab.add(segs, &f.d_segs);
        slab.add(msgs, &f.d_msgs);
        slab.add(work, &f.d_work);
    }
    slab.host.reserve((round1 ? segs.size() * sizeof(SrcSeg) + n * sizeof(SegMsg) + work.size() * sizeof(SrcWork) : 0) + lean_units.size() * sizeof(LeanUnit) +
                      rem.size() * sizeof(DevSrcDesc) + ramp_jobs.size() * sizeof(RampJob) + 8 * 256);     // (one allocation, nothing copied twice)
    if (lean) slab.add(lean_units, &f.d_lean_units);
    slab.add(rem, &f.d_rem);
    slab.add(std::vector<uint32_t>(2, 0u), &f.d_counter);            // {units claimed, waves finished}: zero between launches
    slab.add(dev_jobs, &f.d_ramp_jobs);
    slab.reserve((plane_entries ? plane_entries : 8) * sizeof(uint16_t), &f.d_planes);
    mark("slab");
    int err = slab.upload(ctx, &f.d_slab);
    if (err != OHGPU_OK) { free_src_fast(ctx, b); return err; }
    mark("upload");
    {   // the planes: preset to "no ramp", then RampApplicator's multiplier for every frame of a ramped message (device)
        hipStream_t s0 = ctx ? ctx->stream : nullptr;
        hipError_t e = hipMemsetAsync(f.d_planes, 0xff, (plane_entries ? plane_entries : 8) * sizeof(uint16_t), s0);
        if (e == hipSuccess && ctx) e = launch_ramp_planes(ctx, f.d_ramp_jobs, (uint32_t)ramp_jobs.size(), f.d_planes, s0);
        // (not waited for here: a run of the batch waits for this event on its own stream, on the device)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f.planes_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(f.planes_ready, s0);
        if (e != hipSuccess) { free_src_fast(ctx, b); return set_error(OHGPU_ERR_DEVICE, "ramp planes: %s", hipGetErrorString(e)); }
    }
    mark("planes");
#ifdef OHGPU_PLAN_TIMING
    for (size_t k = 1; k < tps.size(); k++)
        fprintf(stderr, "[plan timing]   %s %.2f ms\n", tps[k].first, std::chrono::duration<double, std::milli>(tps[k].second - tps[k - 1].second).count());
    fprintf(stderr, "[plan timing]   slab %zu bytes, %zu units, %zu ramp jobs, %zu plane entries\n", slab.host.size(), lean_units.size(), ramp_jobs.size(), plane_entries);
#endif
    f.enabled = true;
    // (what ohgpu_src_batch_set_ramps and ohgpu_src_batch_advance go by: small next to the messages -- a job per ramped message and
    // unit, a piece per block-unaligned message end)
    f.host_jobs = dev_jobs;
    f.job_msg.resize(ramp_jobs.size());
    for (size_t k = 0; k < ramp_jobs.size(); k++) f.job_msg[k] = ramp_jobs[k].msg;
    f.host_rem = rem;
    f.rem_msg = rem_msg;
    f.plane_entries = plane_entries;
    f.stream_start = stream_start;
    f.T = T;
    f.n_work = (uint32_t)work.size();
    f.n_lean = (uint32_t)lean_units.size();
    for (const LeanUnit& u : lean_units) f.n_long += ((u.flags >> 8) & 0xffu) > 1 ? 1u : 0u;
    f.n_rem = rem.size();
    f.coef_lds_bytes = coef_lds;
    f.wave_lds_bytes = wave_lds;
    f.max_waves = max_waves;
    f.ring_bytes = ring;
    f.lean = lean;
    f.lean_only = lean_only;
    f.lean_halfband = lean_hb;
    f.wg_only = wg_only || wg_long_units;
    f.lean_coef_lds_bytes = lean_coef;
    f.lean_wave_lds_bytes = lean_wave_lds;
    f.plane_stride = 16;                                              // SrcWork::plane counts 16-byte pieces
    f.lean_max_waves = lean_max_waves;
    f.mfma = mfma;
    f.mfma_wg = mfma_wg;
    f.mfma_wg_halfband = mfma_wg && wg_hb;
    f.wg_unit_rows = mfma_wg ? rows : 0u;
    f.d_mf_amat = (mfma || mfma_wg) ? flt->d_mf_amat : nullptr;
    f.d_mf_steps = (mfma || mfma_wg) ? flt->d_mf_steps : nullptr;
    f.fast_out_frames = fast_frames;
    SrcFastParams& p = f.params;
    memset(&p, 0, sizeof(p));
    p.segs = (const SrcSeg*)f.d_segs;
    p.msgs = (const SegMsg*)f.d_msgs;
    p.work = (const SrcWork*)f.d_work;
    p.coef = flt->d_coef;
    p.src_arena_bytes = b->src_arena_bytes;
    p.L = L; p.M = M; p.L_blk = L_blk; p.M_blk = M_blk;
    p.channels = ch; p.sb = sb; p.db = db;              // (a planar batch: launch_src_lean picks its instantiation by b->src_planar)
    p.src_le = src_le;
    p.dst_le = dst_le;
    return OHGPU_OK;
}

}  // namespace ohgpu

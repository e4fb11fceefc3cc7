// src_plan.cpp -- host-side planning for the block resampler kernel: group output messages into
// contiguous stream segments, cut the segments into phase-aligned blocks, hand everything that is not a whole
// block (segment heads/tails, segments that break an alignment rule) to the generic kernel as message pieces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <utility>
#include <vector>

#include "ohgpu_internal.h"
#include "src_block_common.h"

namespace ohgpu {

void free_src_fast(ohgpu_batch* b)
{
    SrcFastPlan& f = b->fast;
    if (f.d_slab) (void)hipFree(f.d_slab);                            // (every d_* below points into it)
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
    int upload(void** slab)
    {
        *slab = nullptr;
        if (host.empty()) return OHGPU_OK;
        hipError_t e = hipMalloc(slab, host.size());
        if (e == hipSuccess) e = hipMemcpy(*slab, host.data(), host.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE,
                                              "block plan upload: %s", hipGetErrorString(e));
        for (auto& a : at) *a.first = (uint8_t*)*slab + a.second;
        return OHGPU_OK;
    }
};

// RampApplicator's multiplier for frame i of a message of n frames (OpenHome/Media/Pipeline/Msg.cpp:826-837): the ramp value
// moves from `start` towards `end` by C integer division (truncation toward zero, the numerator may be negative), is cast
// to TUint16, and indexes RampArray.h's 512 entries through (kMax - ramp + 16) >> 5, limited to the last entry.
static uint16_t ramp_multiplier(const uint16_t table[512], uint32_t start, uint32_t end, uint32_t i, uint32_t n)
{
    uint32_t ramp = start;
    if (n != 1) {
        const int32_t total = (int32_t)start - (int32_t)end;
        const int32_t prod = (int32_t)i * total;                       // TInt arithmetic (validated: n <= 131071, |total| <= 16384)
        ramp = start - (uint32_t)(prod / (int32_t)(n - 1));
    }
    ramp &= 0xffffu;
    const uint32_t idx = (16384u - ramp + 16u) >> 5;
    return table[idx < 511u ? idx : 511u];
}

// piece [m_lo, m_hi) of message d (device form dv) for the generic kernel
static DevSrcDesc make_piece(const ohgpu_src_msg_desc& d, const DevSrcDesc& dv, uint64_t m_lo, uint64_t m_hi,
                             uint64_t L, uint64_t M)
{
    DevSrcDesc o = dv;
    const uint64_t t_first = m_lo * M;
    o.in_rel0 = (int64_t)(t_first / L) - (int64_t)d.src_frame0;
    o.phase0 = (uint32_t)(t_first % L);
    o.n_frames = (uint32_t)(m_hi - m_lo);
    o.ramp_i0 = (uint32_t)(m_lo - d.out_frame0);
    o.ramp_n = d.n_frames;
    o.dst_offset = d.dst_offset + (m_lo - d.out_frame0) * (uint64_t)d.channels * (d.dst_bits / 8);
    return o;
}

int plan_src_fast(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_src_msg_desc* descs, size_t n,
                  const std::vector<DevSrcDesc>& dev)
{
    (void)ctx;
    SrcFastPlan& f = b->fast;
    f = SrcFastPlan();
    if (n == 0 || !b->uniform) return OHGPU_OK;
    const ohgpu_src* flt = b->src;
    const uint32_t L = flt->L, M = flt->M, T = flt->T;
    const uint32_t ch = b->channels, sb = b->src_bits / 8, db = b->dst_bits / 8;
    const uint32_t src_le = (b->src_endian == OHGPU_ENDIAN_LITTLE && sb > 1) ? 1 : 0;
    const uint32_t dst_le = (b->dst_endian == OHGPU_ENDIAN_LITTLE) ? 1 : 0;
    // a planar source (OHGPU_FLAG_SRC_PLANAR32) is, per channel, a stream of 4-byte frames: the lean kernel alone reads it
    // (stereo to S24); a packed 32-bit source stays on the generic kernel
    const bool planar = b->src_planar;
    if (planar ? !(T == 32 && ch == 2 && sb != 4 && db == 3) : (sb == 4 || !src_block_supported(T, ch, sb, src_le, db, dst_le))) return OHGPU_OK;
    const uint32_t sb_geo = planar ? 3u : sb;                           // (round 1's geometry: only its rows and ring are used)
    const uint32_t sb_lean = planar ? 4u : sb;                          // (LeanGeom: 4 = planar)
    const uint32_t fb_src = planar ? 4u : ch * sb, fb_dst = ch * db;    // (planar: a plane's frame)
    // the ring is drained every four advances: #{j : a <= floor(j*M/L) < a+4} <= ceil(4L/M) outputs arrive in between
    const uint32_t out_per_drain = (4 * L + M - 1) / M;
    uint32_t rows = 0, ring = 0, coef_lds = 0, wave_lds = 0, max_waves = 0;
    if (!src_block_geometry(L, T, ch, sb_geo, db, out_per_drain, &rows, &ring, &coef_lds, &wave_lds, &max_waves)) return OHGPU_OK;
    // the lean kernel (round 2): same blocks, rows and ring; its rounding bias needs sum|c| < 2^29 in every phase
    uint32_t lean_rows = 0, lean_inb = 0, lean_sf = 8, lean_ring = 0, lean_coef = 0, lean_wave_lds = 0, lean_max_waves = 0;
    const bool lean = flt->max_sum_abs < ((int64_t)1 << 29) &&
                      src_lean_geometry(L, T, ch, sb_lean, db, out_per_drain, &lean_rows, &lean_inb, &lean_sf, &lean_ring, &lean_coef, &lean_wave_lds, &lean_max_waves) &&
                      lean_rows == rows && lean_ring == ring;
    if (planar && !lean) return OHGPU_OK;
    // a block: whole phase periods (multiple of L), at least 128 outputs, and a whole number of 64-byte output lines
    uint32_t min_blk = 128;
#ifdef OHGPU_DIAG
    if (const char* e = getenv("OHGPU_DIAG_MIN_BLOCK")) min_blk = (uint32_t)atoi(e);     // (diagnostic builds: longer blocks per lane)
#endif
    uint32_t L_blk = L * ((min_blk + L - 1) / L);
    {
        uint32_t k = 1;
        while (k <= 64 && ((uint64_t)L_blk * k * fb_dst) % 64 != 0) k++;
        if (k > 64) return OHGPU_OK;
        L_blk *= k;
    }
    const uint64_t M_blk64 = (uint64_t)L_blk * M / L;
    if (M_blk64 + T > 32000 || M_blk64 < T) return OHGPU_OK;    // a block is at least one filter length of input
    const uint32_t M_blk = (uint32_t)M_blk64;

    // order messages by (stream, output position); a stream is identified by where its absolute frame 0 lives
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    auto src_base_of = [&](const ohgpu_src_msg_desc& d) { return (int64_t)d.src_offset - (int64_t)(d.src_frame0 * fb_src); };
    // (planar: two messages of a stream also agree on the distance between its planes; a 4 GiB reach per unit is the lean kernel's)
    if (planar) {
        for (size_t k = 0; k < n; k++)
            if ((uint64_t)(ch - 1) * descs[k].src_plane_stride + (uint64_t)(rows + 1) * (M_blk + T) * 4 + 4096 >= (1ull << 32)) return OHGPU_OK;
    }
    auto dst_base_of = [&](const ohgpu_src_msg_desc& d) { return (int64_t)d.dst_offset - (int64_t)(d.out_frame0 * fb_dst); };
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const int64_t sx = src_base_of(descs[x]), sy = src_base_of(descs[y]);
        if (sx != sy) return sx < sy;
        if (descs[x].src_plane_stride != descs[y].src_plane_stride) return descs[x].src_plane_stride < descs[y].src_plane_stride;
        const int64_t dx = dst_base_of(descs[x]), dy = dst_base_of(descs[y]);
        if (dx != dy) return dx < dy;
        return descs[x].out_frame0 < descs[y].out_frame0;
    });

    std::vector<SrcSeg> segs;
    std::vector<uint32_t> seg_plane_stride;             // planar batches: bytes between a segment's planes
    std::vector<SegMsg> msgs;
    std::vector<SrcWork> work;
    std::vector<DevSrcDesc> rem;
    uint64_t fast_frames = 0;
    // lean kernel: one plane of multipliers per ramped unit -- rows * L_blk entries (uint16, 0xffff = no ramp on that frame)
    // (the kernel loads eight entries at a time; a row is a whole number of loads when L_blk is a multiple of 8, else the slack covers the last one)
    uint16_t ramp_table[512];
    build_ramp_table(ramp_table);
    std::vector<uint16_t> planes;

    size_t i = 0;
    while (i < n) {
        // grow a run of messages that tile a contiguous output range of one stream
        size_t e = i + 1;
        const ohgpu_src_msg_desc& d0 = descs[order[i]];
        const int64_t sbase = src_base_of(d0), dbase = dst_base_of(d0);
        uint64_t next_out = d0.out_frame0 + d0.n_frames;
        bool zero_len = d0.n_frames == 0;
        while (!zero_len && e < n) {
            const ohgpu_src_msg_desc& d = descs[order[e]];
            if (d.n_frames == 0 || src_base_of(d) != sbase || dst_base_of(d) != dbase || d.out_frame0 != next_out ||
                d.src_plane_stride != d0.src_plane_stride) break;
            next_out += d.n_frames;
            e++;
        }
        const uint64_t m_begin = d0.out_frame0, m_end = next_out;
        uint64_t blk_lo = (m_begin + L_blk - 1) / L_blk, blk_hi = m_end / L_blk;
        // (a stream whose output does not start on a 64-byte boundary is written with unaligned 16-byte stores: they run at
        // the aligned rate -- tools/micro/unaligned_store.hip -- only the lines are then no longer whole)
        bool fast_ok = !zero_len && blk_hi > blk_lo;
        if (fast_ok) {
            // every whole block's history must be present in the windows the caller declared (they were validated
            // per message; the block reads nothing a message of the block does not itself need)
            const uint32_t msg_begin = (uint32_t)msgs.size();
            for (size_t k = i; k < e; k++) {
                const ohgpu_src_msg_desc& d = descs[order[k]];
                SegMsg sm;
                memset(&sm, 0, sizeof(sm));
                sm.out0 = d.out_frame0; sm.n = d.n_frames; sm.ramp_start = d.ramp_start; sm.ramp_end = d.ramp_end; sm.flags = d.flags;
                if (d.flags & OHGPU_FLAG_RAMP) {            // RampApplicator divides by n - 1 per frame (Msg.cpp:835): exact multiplier instead
                    uint32_t sh = 0;
                    magic_u31(d.n_frames > 1 ? d.n_frames - 1 : 1, &sm.m_n1, &sh);
                    sm.s_n1 = (uint8_t)sh;
                }
                msgs.push_back(sm);
            }
            SrcSeg sg;
            sg.src_base = sbase; sg.dst_base = dbase; sg.msg_begin = msg_begin; sg.msg_end = (uint32_t)msgs.size();
            const uint32_t seg_index = (uint32_t)segs.size();
            segs.push_back(sg);
            seg_plane_stride.push_back((uint32_t)d0.src_plane_stride);
            uint32_t mi = msg_begin;                      // message that holds the unit's first output frame
            for (uint64_t bk = blk_lo; bk < blk_hi; bk += rows) {
                SrcWork w;
                w.first_block = bk; w.seg = seg_index; w.n_blocks = (uint32_t)std::min<uint64_t>(rows, blk_hi - bk);
                while (mi + 1 < sg.msg_end && msgs[mi + 1].out0 <= bk * L_blk) mi++;
                w.msg_first = mi;
                // cost class, for the order below: a wave that meets a ramped message goes through the per-output ramp path
                // for all of its lanes, which makes such a unit two to three times as long as a plain one
                const uint64_t u_lo = bk * L_blk, u_hi = (bk + w.n_blocks) * L_blk;
                bool ramped = false;
                for (uint32_t m = mi; m < sg.msg_end && msgs[m].out0 < u_hi && !ramped; m++)
                    ramped = (msgs[m].flags & OHGPU_FLAG_RAMP) && msgs[m].out0 + msgs[m].n > u_lo;
                w.flags = ramped ? kWorkRamped : 0u;
                w.plane = 0; w.pad = 0;
                if (ramped && lean) {
                    // (as many rows as the unit has blocks -- a live pipeline's batch of one message per stream has one -- in
                    // whole 16-byte pieces; the kernel addresses a plane as planes + plane * plane_stride with a stride of 16)
                    const size_t per = (((size_t)w.n_blocks * L_blk + 8) + 7) & ~(size_t)7;
                    if (planes.size() / 8 + per / 8 > 0xffffffffull) return OHGPU_OK;
                    w.plane = (uint32_t)(planes.size() / 8);
                    const size_t at = planes.size();
                    planes.resize(at + per, 0xffffu);
                    uint16_t* pl = planes.data() + at;
                    for (uint32_t m = mi; m < sg.msg_end && msgs[m].out0 < u_hi; m++) {
                        const SegMsg& sm = msgs[m];
                        if (!(sm.flags & OHGPU_FLAG_RAMP)) continue;
                        const uint64_t lo = std::max<uint64_t>(sm.out0, u_lo), hi = std::min<uint64_t>(sm.out0 + sm.n, u_hi);
                        for (uint64_t fr = lo; fr < hi; fr++)
                            pl[fr - u_lo] = ramp_multiplier(ramp_table, sm.ramp_start, sm.ramp_end, (uint32_t)(fr - sm.out0), sm.n);
                    }
                }
                // the lean kernel's staging moves, per stage q, the aligned 16-byte pieces that hold each row's eight frames:
                // does every one of them lie inside the arena?  (Only a unit at an end of the arena can fail.)
                {
                    const int64_t total = (int64_t)M_blk + T, n_stages = (total + lean_sf - 1) / lean_sf;
                    for (uint32_t pc = 0; pc < (planar ? ch : 1u); pc++) {          // (planar: every channel's plane is staged on its own)
                        const int64_t g_first = sbase + (int64_t)(pc * d0.src_plane_stride) + ((int64_t)(bk * M_blk) - (int64_t)T) * fb_src;
                        const int64_t g_last = g_first + (int64_t)(w.n_blocks - 1) * M_blk * fb_src;
                        const int64_t lo = g_first - (g_first & 15);
                        const int64_t hi = g_last - (g_last & 15) + 16 * (((g_last & 15) + lean_sf * fb_src + 15) >> 4) + (n_stages - 1) * lean_sf * fb_src;
                        if (g_first < 0 || lo < 0 || (uint64_t)hi > b->src_arena_bytes) w.flags |= kWorkChecked;
                    }
                }
                work.push_back(w);
            }
            fast_frames += (blk_hi - blk_lo) * L_blk;
        } else {
            blk_lo = blk_hi = 0;   // everything goes to the generic kernel
        }
        const uint64_t fast_lo = fast_ok ? blk_lo * L_blk : m_end, fast_hi = fast_ok ? blk_hi * L_blk : m_end;
        for (size_t k = i; k < e; k++) {
            const ohgpu_src_msg_desc& d = descs[order[k]];
            if (d.n_frames == 0) continue;
            const uint64_t lo = d.out_frame0, hi = d.out_frame0 + d.n_frames;
            if (!fast_ok) { rem.push_back(make_piece(d, dev[order[k]], lo, hi, L, M)); continue; }
            if (lo < fast_lo) rem.push_back(make_piece(d, dev[order[k]], lo, std::min(hi, fast_lo), L, M));
            if (hi > fast_hi) rem.push_back(make_piece(d, dev[order[k]], std::max(lo, fast_hi), hi, L, M));
        }
        i = e;
    }
    if (work.empty()) return OHGPU_OK;
    if (!planes.empty()) planes.resize(planes.size() + (size_t)rows * L_blk + 8, 0xffffu);   // (lanes without a block read their row's place too)
    // Longest first: the waves claim units in this order, and the kernel ends when the last unit does.  With the ramped
    // units where the streams put them (each stream's fade-out is its last units) the launch ended on a few long units
    // with most of the chip idle.
    std::stable_sort(work.begin(), work.end(), [](const SrcWork& x, const SrcWork& y) {     // ramped first, partly filled units last
        return ((x.flags & kWorkRamped) ? 3u : 1u) * x.n_blocks > ((y.flags & kWorkRamped) ? 3u : 1u) * y.n_blocks;
    });

    std::vector<LeanUnit> lean_units;
    if (lean) {
        lean_units.reserve(work.size());
        for (const SrcWork& w : work) {
            LeanUnit u;
            u.src_row0 = segs[w.seg].src_base + ((int64_t)(w.first_block * M_blk) - (int64_t)T) * fb_src;
            u.dst_row0 = segs[w.seg].dst_base + (int64_t)(w.first_block * L_blk) * fb_dst;
            u.n_blocks = w.n_blocks;
            u.flags = w.flags | (w.first_block == 0 ? (uint32_t)kWorkFirst : 0u);
            u.plane = w.plane;
            u.src_plane_stride = planar ? seg_plane_stride[w.seg] : 0u;
            lean_units.push_back(u);
        }
    }

    Slab slab;
    slab.add(segs, &f.d_segs);
    slab.add(msgs, &f.d_msgs);
    slab.add(work, &f.d_work);
    if (lean) slab.add(lean_units, &f.d_lean_units);
    slab.add(rem, &f.d_rem);
    slab.add(planes.empty() ? std::vector<uint16_t>(4, 0xffffu) : planes, &f.d_planes);
    slab.add(std::vector<uint32_t>(2, 0u), &f.d_counter);            // {units claimed, waves finished}: zero between launches
    int err = slab.upload(&f.d_slab);
    if (err != OHGPU_OK) { free_src_fast(b); return err; }
    f.enabled = true;
    f.T = T;
    f.n_work = (uint32_t)work.size();
    f.n_rem = rem.size();
    f.coef_lds_bytes = coef_lds;
    f.wave_lds_bytes = wave_lds;
    f.max_waves = max_waves;
    f.ring_bytes = ring;
    f.lean = lean;
    f.lean_coef_lds_bytes = lean_coef;
    f.lean_wave_lds_bytes = lean_wave_lds;
    f.plane_stride = 16;                                              // SrcWork::plane counts 16-byte pieces
    f.lean_max_waves = lean_max_waves;
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

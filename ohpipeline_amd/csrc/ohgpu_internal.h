// ohgpu_internal.h -- host-side structures behind the opaque handles of include/ohgpu.h.
#pragma once

#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/ohgpu.h"

namespace ohgpu {

// Device-side form of one piece of a resampled output message (56 bytes): everything 64-bit that can be
// precomputed on the host has been.  in_rel0 = n0(first output) - src_frame0 (negative only at a stream
// start, where frames with a negative absolute index read as zero); phase0 = (first output * M) % L.
// A piece is a whole message, or the part of one that the block kernel leaves to the generic kernel;
// ramp_i0 / ramp_n place it inside its message for RampApplicator's (iLoopCount, iNumSamples).
struct DevSrcDesc {
    uint64_t src_offset;
    uint64_t dst_offset;
    int64_t  in_rel0;
    uint32_t phase0;
    uint32_t n_frames;
    uint32_t ramp_i0;
    uint32_t ramp_n;
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint8_t  channels;
    uint8_t  src_bits;
    uint8_t  src_endian;
    uint8_t  dst_bits;
    uint8_t  dst_endian;
    uint8_t  flags;
    uint8_t  pad[2];
    uint32_t plane_frames;    // OHGPU_FLAG_SRC_PLANAR32: frames (4-byte units) between the channels' planes
};
static_assert(sizeof(DevSrcDesc) == 56, "DevSrcDesc layout");
static_assert(sizeof(ohgpu_msg_desc) == 32, "ohgpu_msg_desc layout");
static_assert(sizeof(ohgpu_src_msg_desc) == 64, "ohgpu_src_msg_desc layout");
static_assert(sizeof(ohgpu_fmt_desc) == 48, "ohgpu_fmt_desc layout");

// ---- block ("fast") resampler plan: contiguous runs of output messages cut into phase-aligned blocks ----
struct SrcSeg {               // one contiguous run of output messages of one stream
    int64_t  src_base;        // byte offset (in the source arena) of the stream's absolute input frame 0
    int64_t  dst_base;        // byte offset (in the destination arena) of the stream's absolute output frame 0
    uint32_t msg_begin;       // [msg_begin, msg_end) in the SegMsg array, ordered by out0
    uint32_t msg_end;
};
struct SegMsg {               // ramp parameters of one output message, 24 bytes
    uint64_t out0;            // absolute index of its first output frame
    uint32_t n;               // frames
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint8_t  flags;
    uint8_t  s_n1;            // x / (n - 1) == umulhi(x, m_n1) >> s_n1 for x < 2^31 (m_n1 == 0: n - 1 <= 1): the ramp's division
    uint8_t  pad[2];
    uint32_t m_n1;
};
static_assert(sizeof(SegMsg) == 24, "SegMsg");
struct SrcWork {              // one workgroup's share: up to `rows` consecutive blocks of one segment
    uint64_t first_block;     // absolute block index (block b covers outputs [b*L_blk, (b+1)*L_blk))
    uint32_t seg;
    uint32_t n_blocks;
    uint32_t msg_first;       // index (in the SegMsg array) of the message that holds the unit's first output frame
    uint32_t flags;           // kWorkRamped | kWorkChecked (src_block_common.h)
    uint32_t plane;           // a ramped unit's multiplier plane (lean kernel): its index in SrcFastPlan::d_planes
    uint32_t pad;
};
struct LeanUnit {             // the lean kernel's own view of a unit (same order as the SrcWork array): ONE 32-byte scalar load,
                              // nothing to look up behind it -- a unit's set-up used to wait for work[], then for segs[work.seg]
    int64_t  src_row0;        // source arena offset of row 0's input frame at advance -T: seg.src_base + (first_block * M_blk - T) * fb_src
    int64_t  dst_row0;        // destination arena offset of row 0's first output byte: seg.dst_base + first_block * L_blk * fb_dst
    uint32_t n_blocks;
    uint32_t flags;           // kWorkRamped | kWorkChecked | kWorkFirst; bits 8..15: blocks per ROW (a row = that many consecutive blocks)
    uint32_t plane;
    uint32_t src_plane_stride; // planar source: bytes between the channels' planes (the planner keeps such a batch off this kernel
                               // unless every plane of a unit is within 4 GiB of its first)
};
struct RampJob {              // ramp_plane_kernel: frames [i0, i0 + count) of one ramped message go to entries [plane_entry, + count) of the planes
    uint64_t plane_entry;     // index of the first uint16 entry
    uint32_t i0, count;
    uint32_t n;               // the message's frames (RampApplicator's iNumSamples)
    uint32_t m_n1;            // x / (n - 1) == umulhi(x, m_n1) >> s_n1 for x < 2^31 (m_n1 == 0: n - 1 <= 1)
    uint16_t ramp_start, ramp_end;
    uint8_t  s_n1;
    uint8_t  pad[3];
};
static_assert(sizeof(SegMsg) == 24 && sizeof(SrcWork) == 32 && sizeof(SrcSeg) == 24 && sizeof(LeanUnit) == 32 && sizeof(RampJob) == 32, "plan layouts");
struct MfStep {               // src_mfma_kernel: one step = 16 consecutive output frames of a row (the same for every row: rows start at phase 0)
    uint32_t aoff[16];        // output m's A row: byte offset into a digit's table, phase * 96 + (31 + k0 - n0(m))
    uint32_t b0[16], b1[16], b2[16];   // its accumulators' initial values: bits 0..15 and (signed) bits 16.. of 32896 * sum(c[phase]) + 2^27 -- below 2^28, so
                              // that class 2's accumulator holds them beside its own 2^22 -- and zero (b2: a third read per tile until round 4's end)
    uint32_t kc;              // the step's window is input chunks kc .. kc + 3 (a chunk = 16 frames; frame 0 of chunk 0 = the row's frame -32)
    uint32_t pad[7];
};
static_assert(sizeof(MfStep) == 288, "MfStep");

struct SrcFastParams {        // kernel argument block
    const SrcSeg*  segs;
    const SegMsg*  msgs;
    const SrcWork* work;
    const double*  coef;      // [L][T]
    const uint16_t* ramp_table;
    const uint8_t* src;
    uint8_t*       dst;
    uint64_t src_arena_bytes;
    uint32_t L, M;
    uint32_t L_blk, M_blk;    // outputs / inputs per block
    uint32_t channels, sb, db; // the batch's layout: selects the kernel instantiation
    uint32_t src_le, dst_le;
};

struct SrcFastPlan {
    bool     enabled = false;
    uint32_t T = 0;               // taps per phase
    SrcFastParams params{};
    uint32_t n_work = 0;
    uint32_t n_long = 0;          // ... of which units whose rows are several blocks long
    uint32_t n_lean = 0;          // LeanUnit count (a lean unit may hold several blocks per row: it need not equal n_work)
    uint32_t coef_lds_bytes = 0;  // the coefficient table's share of a workgroup's LDS
    uint32_t wave_lds_bytes = 0;  // per wave: input stages, message table, output ring
    uint32_t max_waves = 0;       // waves per workgroup the LDS allows (<= 12)
    uint32_t ring_bytes = 0;      // bytes of packed output a block row's LDS ring holds
    bool     lean = false;        // the batch runs on src_lean_kernel (round 2) rather than src_block_kernel
    bool     lean_only = false;   // ... and round 1's kernel has no instantiation for its layout (variant 2 then runs the lean kernel too)
    bool     lean_halfband = false;   // ... in its half-band form (the filter is one AND the layout has that instantiation): LDS sizing and dispatch agree on this
    bool     wg_only = false;     // ... or no block kernel but src_mfma_wg_kernel has one: any variant that asks for another kernel gets the generic one
    uint32_t lean_coef_lds_bytes = 0, lean_wave_lds_bytes = 0, lean_max_waves = 0;
    bool     mfma = false;        // ... and its layout is one src_mfma_kernel (round 4) serves: same units and planes, the filter's digit tables
    bool     mfma_wg = false;     // ... as src_mfma_wg_kernel cuts them: one unit per workgroup (rows of ONE block; the units whose input image leaves the arena -- kWorkEdge, in front of the list -- run on it too, through its checked loads)
    bool     mfma_wg_halfband = false;   // ... in its half-band form (the filter's tables are build_mfma_halfband's)
    uint32_t wg_unit_rows = 0;    // ... and the rows its units were cut to (WgGeom::kUnitRows: the launch checks)
    const void* d_mf_amat = nullptr;   // (owned by the ohgpu_src)
    const void* d_mf_steps = nullptr;
    void*    d_planes = nullptr;  // uint16: RampApplicator's multiplier per output frame of every ramped unit, [blocks of the unit][L_blk]
    hipEvent_t planes_ready = nullptr;   // recorded on the context's stream behind the kernel that fills them: a run on any stream waits for it (on the device)
    uint32_t plane_stride = 0;    // the unit of SrcWork::plane / LeanUnit::plane in bytes (16: planes are as long as their units)
    void*    d_slab = nullptr;    // the one allocation the arrays below live in
    void*    d_segs = nullptr;
    void*    d_msgs = nullptr;
    void*    d_work = nullptr;
    void*    d_lean_units = nullptr;   // LeanUnit [n_work], when `lean`
    void*    d_counter = nullptr; // uint32[2]: units claimed / waves finished by the running launch; the kernel's last wave zeroes them
    void*    d_ramp_jobs = nullptr; // RampJob[]: what ramp_plane_kernel filled the planes from (kept for the batch's lifetime)
    void*    d_rem = nullptr;     // DevSrcDesc[] the generic kernel finishes (block-unaligned heads and tails)
    size_t   n_rem = 0;
    uint64_t fast_out_frames = 0;
    // host copies of what carries the messages' ramp endpoints on the device, and whose they are (the caller's message index):
    // ohgpu_src_batch_set_ramps rewrites them; `stream_start`: some message's filter window reaches in front of its stream's first
    // frame (such a batch cannot be advanced: ohgpu_src_batch_advance)
    std::vector<RampJob> host_jobs;
    std::vector<uint32_t> job_msg;
    std::vector<DevSrcDesc> host_rem;
    std::vector<uint32_t> rem_msg;
    size_t   plane_entries = 0;
    bool     stream_start = false;
    uint64_t advanced_blocks = 0;       // ohgpu_src_batch_advance: blocks added to every message's position since creation
};

int plan_thread_cap();     // ohgpu_set_plan_threads (0: no cap)
// How many threads a host loop over n independent items is worth (at least `per_thread` items each, at most 16 and what the host grants the process)
unsigned usable_cpus();    // csrc/ohgpu_api.hip: the affinity mask's CPUs, capped by the container's CPU quota
inline unsigned plan_threads(size_t n, size_t per_thread)
{
    const unsigned hw = usable_cpus();
    size_t t = n / (per_thread ? per_thread : 1);
    if (t < 1) t = 1;
    if (t > 16) t = 16;
    if (hw && t > hw) t = hw;
    if (ohgpu::plan_thread_cap() > 0 && t > (size_t)ohgpu::plan_thread_cap()) t = (size_t)ohgpu::plan_thread_cap();
    return (unsigned)t;
}
// f(range, lo, hi) over [0, n) cut into n_ranges contiguous ranges, run by up to `threads` threads of the library's planning pool (the
// caller's among them), each claiming the next range when it has done one.  The pool's threads are started on first use and sleep
// between jobs: starting sixteen threads per pass cost more than the passes of a half-million-message plan.
void run_on_pool(unsigned n_ranges, unsigned threads, void (*job)(void* arg, unsigned t), void* arg);     // csrc/ohgpu_api.hip
template <typename F>
inline void parallel_ranges(size_t n, unsigned n_ranges, unsigned threads, F&& f)
{
    if (n_ranges <= 1) { f(0u, (size_t)0, n); return; }
    struct Ctx { F* f; size_t n; unsigned n_ranges; } c{&f, n, n_ranges};
    run_on_pool(n_ranges, threads, [](void* a, unsigned t) {
        Ctx* c = (Ctx*)a;
        (*c->f)(t, c->n * t / c->n_ranges, c->n * (t + 1) / c->n_ranges);
    }, &c);
}
template <typename F>
inline void parallel_ranges(size_t n, unsigned n_thr, F&& f) { parallel_ranges(n, n_thr, n_thr, static_cast<F&&>(f)); }

// x / d == umulhi(x, m) >> s for every x < 2^31 (d >= 2; m == 0 stands for d == 1): with 2^(l-1) < d <= 2^l and
// m = floor(2^(31+l) / d) + 1 the error term x * (m * d - 2^(31+l)) stays below 2^(31+l).  Host side of the line kernels.
inline void magic_u31(uint32_t d, uint32_t* m, uint32_t* s)
{
    if (d <= 1) { *m = 0; *s = 0; return; }
    const uint32_t l = 32u - (uint32_t)__builtin_clz(d - 1u);          // 2^(l-1) < d <= 2^l
    *m = (uint32_t)(((1ull << (31 + l)) / d) + 1);
    *s = l - 1;
}

// ---- line kernel of the PCM message path (csrc/pcm_line_kernel.hip) ----
struct PcmChunk {             // one wave's share of a message: subsamples [q0, q0 + nq); 64 bytes = one scalar load
    // ---- the first 16 bytes are what it takes to START a chunk (issue its loads): the uniform path fetches them for the
    // next trip's chunks ahead of time and the whole record only once the audio is on its way (PcmChunkHead) ----
    uint64_t src_off;             // byte offset of the chunk's first source byte in the arena
    uint32_t nq;
    uint8_t  channels, sb, db, flags;   // bytes per subsample; kChunk* bits
    uint64_t dst_off;             // byte offset of the chunk's first destination byte
    uint32_t q0;
    uint32_t n_frames;            // of the whole message (ramp)
    uint16_t ramp_start, ramp_end;
    uint32_t attenuation;
    uint32_t m_ch, m_n1;          // x / d == umulhi(x, m) >> s for x < 2^31 (m == 0: d == 1); d = channels, n_frames - 1
    uint8_t  s_ch, s_n1;
    uint8_t  prefix_bytes;        // bytes the chunk's wave copies from the batch's prefix blob to just before dst_off (0: none) --
    uint8_t  pad8;                //   a Songcast frame's header, written with the frame's first audio (csrc/ohm_frame_kernel.hip)
    uint32_t plain_sel;           // v_perm_b32 selector of the plain path: source bytes -> destination bytes in memory order
    uint32_t prefix_off;          // of those bytes in the blob (a multiple of 4; entries are padded to whole dwords)
    uint32_t pad;
};
struct PcmChunkHead { uint64_t src_off; uint32_t nq; uint8_t channels, sb, db, flags; };
static_assert(sizeof(PcmChunkHead) == 16, "the head of a chunk record");
static_assert(sizeof(PcmChunk) == 64, "chunk record = one 64-byte scalar load");
enum { kChunkRamp = 1, kChunkSilence = 2, kChunkZeroLsb = 4, kChunkSrcLe = 8, kChunkDstLe = 16 };
constexpr uint32_t kLineLists = 10;   // chunk lists of a plan: [0] the general path, [1 + (sb - 2) * 3 + (db - 2)] the 16/24/32-bit layouts
struct PcmLinePlan {
    bool     enabled = false;
    bool     prefixed = false;    // chunks carry prefixes (d_prefix): only this kernel writes them
    uint32_t n_chunks = 0;
    uint32_t list_first[kLineLists] = {}, list_count[kLineLists] = {};   // d_chunks[list_first[k], + list_count[k]): one launch each
    uint8_t  list_heavy[kLineLists] = {};   // ... and the share of each list's subsamples, in percent, that is ramped or attenuated (the launch's occupancy)
    void*    d_chunks = nullptr;
    void*    d_prefix = nullptr;  // the prefix blob
};
struct MsgPrefix { uint32_t off; uint32_t bytes; };   // per message: [off, off + bytes) of the blob goes right before its destination (bytes <= 255; 0: none)

// ---- line kernel of the layout-changing processors (csrc/fmt_line_kernel.hip) ----
struct FmtChunk {             // 64 bytes = one scalar load
    uint64_t src_off, dst_off;    // first source byte of run 0 / first destination byte
    uint64_t run_src_stride;      // run r starts at src_off + r * run_src_stride (a14: one run per plane)
    uint32_t nq;                  // destination subsamples
    uint32_t run_bytes;           // source bytes per run
    uint16_t n_runs;
    uint16_t run_lds_stride;      // run r is staged run_lds_stride bytes after run r - 1
    uint16_t map_a, map_b, map_c, map_d;   // source subsample of destination subsample q: (q / A) * B + C + (q % A) * D
    uint32_t m_a;                 // q / A == umulhi(q, m_a) >> s_a (m_a == 0: A == 1)
    uint32_t sel;                 // v_perm_b32 selector: source bytes -> destination bytes in memory order
    uint8_t  s_a, sb, db, pad8;
    uint32_t pad[2];
};
static_assert(sizeof(FmtChunk) == 64, "chunk record = one 64-byte scalar load");
struct FmtLinePlan {
    bool     enabled = false;
    uint8_t  group_kind = 0;      // OHGPU_FMT_UNPACK_PLANAR / _FLAC_PACK: a uniform stereo batch on the register-only kernels
    uint8_t  group_bytes = 0;     // its source (a11) / destination (a14) bytes per subsample
    uint32_t n_chunks = 0;
    void*    d_chunks = nullptr;
    uint32_t n_wide = 0;          // a batch of Sender packs that all drop channels: OhmSelRec[n_wide] for ohm_wide_kernel
    void*    d_wide = nullptr;
};

// ---- FlywheelRamper (csrc/flywheel_kernel.hip) ----
struct FlywheelLane { uint32_t req, channel; };      // one lane = one channel of one request
struct FlywheelPlan {
    uint32_t n_lanes = 0, lanes_padded = 0, max_count = 0;
    void*    d_lanes = nullptr;
    void*    d_work = nullptr;    // int16 [3][max_count][lanes_padded]: decimated input, per, pef of Burg's method
};

// ---- Songcast sender frames (csrc/ohm_frame_kernel.hip) ----
struct OhmFrameRec {              // 48 bytes: the 36 per-frame header bytes in wire order, and where they go
    uint64_t dst_off;
    uint32_t w[9];
    uint32_t stream_and_bytes;    // bits 0..23: index of the 64-byte stream record (bytes [0, n) = OhmMsgAudio::GetStreamHeader);
                                  // bits 24..31: the whole header's size, 36 + n
};
struct OhmSelRec {               // 48 bytes: one audible fragment of a stream of more than two channels (ohm_wide_kernel)
    uint64_t src_off, dst_off;
    uint32_t n_frames;
    uint32_t m_n1;                // x / (n_frames - 1) == umulhi(x, m_n1) >> s_n1 for x < 2^31 (m_n1 == 0: n_frames - 1 <= 1)
    uint16_t ramp_start, ramp_end, attenuation, pad16;      // (every field inside an aligned dword: the record is read with scalar loads)
    uint8_t  channels, sb, first_ch, flags;
    uint8_t  s_n1, little;
    uint8_t  prefix_bytes;        // the frame's header: [prefix_off, + prefix_bytes) of the batch's blob, written right before dst_off when
    uint8_t  pad8;                //   the fragment is the frame's first audio (0: it is not)
    uint32_t prefix_off;          // (a multiple of 4)
    uint32_t safe_frames;         // frames [0, safe_frames) may be read with one 8-byte load each (it stays inside the source arena)
};
static_assert(sizeof(OhmSelRec) == 48, "OhmSelRec");
struct OhmPlan {
    ohgpu_batch* direct = nullptr;         // pcm batch: fragments of mono/stereo streams, source -> frames (ramp + depth in one pass, headers as prefixes)
    ohgpu_batch* stage = nullptr;          // pcm batch: silent fragments of wider streams -> scratch
    ohgpu_batch* select_staged = nullptr;  // fmt batch: scratch -> frames
    void*    d_selr = nullptr;             // OhmSelRec[n_selr]: audible fragments of wider streams (ohm_wide_kernel: select, attenuation, ramp, header)
    uint32_t n_selr = 0;
    void*    d_wide_prefix = nullptr;      // their frames' headers
    void*    d_scratch = nullptr;
    void*    d_frames = nullptr;           // OhmFrameRec[n_frames] for ohm_header_kernel: [0, n_unfolded) the headers no audio pass writes,
    void*    d_streams = nullptr;          //   then, up to n_unfolded_generic, those `direct` writes unless the generic kernel runs it
    uint32_t n_frames = 0;
    uint32_t n_unfolded = 0, n_unfolded_generic = 0;
};

enum BatchKind { kBatchPcm = 1, kBatchSrc = 2, kBatchFmt = 3, kBatchFlywheel = 4, kBatchOhm = 5 };

}  // namespace ohgpu

namespace ohgpu {
// Device blocks of the batches' descriptors and plan arrays, kept by the context between batches: a batch that is destroyed gives
// its blocks back (by size class, 256 B << c), the next one of that size takes them -- a steady caller (the StarvationRamper's
// rescue: three batch objects per starving period) allocates on the device once.  Blocks above the largest class go straight to
// hipMalloc / hipFree.  `device_allocs` counts the hipMalloc calls made through it (ohgpu_device_allocations).
struct DevCache {
    static constexpr int kClasses = 15;                // 256 B .. 4 MiB
    std::mutex m;
    std::vector<void*> idle[kClasses];
    std::unordered_map<void*, int> cls;                // every block handed out or idle -> its class (-1: not cached)
    uint64_t device_allocs = 0;
};
}  // namespace ohgpu

namespace ohgpu {
// What the host-buffer calls (ohgpu_*_process_host: a driver thread's period with host memory on its side of the boundary) keep
// from call to call: the two device arenas the audio passes through and a pinned bounce buffer, each grown with headroom when a
// call needs more -- so that a steady caller allocates nothing per period -- and the bytes those calls moved over the link.
struct HostStage {
    void*  d_src = nullptr;    size_t src_cap = 0;
    void*  d_dst = nullptr;    size_t dst_cap = 0;
    void*  h_bounce = nullptr; size_t bounce_cap = 0;
    uint64_t h2d_bytes = 0, d2h_bytes = 0, calls = 0, src_calls = 0;
};
}  // namespace ohgpu

struct ohgpu_ctx {
    ohgpu::DevCache cache;
    ohgpu::HostStage stage;
    int          device;
    hipStream_t  stream;          // the context's own stream (used when the caller passes NULL)
    uint16_t*    d_ramp_table;    // 512 x u16 (RampArray.h:7-74)
    int          variant;         // kernel selection, 0 = best
    int          num_cus;
    char         name[128];
};

struct ohgpu_src {
    uint32_t L, M, T;
    int64_t  max_sum_abs;         // largest sum|c| over the phases, Q28 (the lean kernel's rounding bias needs < 2^29)
    bool     halfband;            // L = 1, M = 2, T = 64 and every odd tap but tap T/2 - 1 zero (host_design.cpp's 2:1 decimator): the
                                  // lean kernel's half-band instantiations multiply by the 33 taps that are not
    double*  d_coef;              // [L][T] exact integer-valued doubles (Q28)
    int32_t* d_coef_q28;          // [L][T] int32
    // src_mfma_kernel's tables (T = 32 filters whose ratio the 16-output tiling holds; null otherwise), made for blocks of
    // mf_L_blk outputs and rows of up to mf_kb_cap blocks
    uint8_t* d_mf_amat = nullptr; // the steps' A operands, lane-linear: [step][4 digits][64 lanes][16 bytes] (build_mfma_images)
    ohgpu::MfStep* d_mf_steps = nullptr;
    uint32_t mf_L_blk = 0, mf_kb_cap = 0;
    bool     mf_halfband = false; // ... in the half-band form: one coefficient image for every step (build_mfma_halfband)
};

namespace ohgpu {
// Host arrays of tens of megabytes that are written once, by several threads: 2 MiB-aligned and advised to the kernel as huge-page
// material, so that filling them costs a page fault per 2 MiB instead of one per 4 KiB (seven thousand of them for the headline's
// half a million descriptors, a third of the time their conversion took).
struct HostFree { void operator()(void* p) const { free(p); } };
inline void* host_alloc_huge(size_t bytes)
{
    const size_t huge = (size_t)2 << 20;
    if (bytes < huge) return malloc(bytes ? bytes : 1);
    const size_t len = (bytes + huge - 1) & ~(huge - 1);
    void* p = aligned_alloc(huge, len);
    if (p) (void)madvise(p, len, MADV_HUGEPAGE);
    return p;
}
}  // namespace ohgpu

struct ohgpu_batch {
    int      kind;
    size_t   n;
    void*    d_descs;             // ohgpu_msg_desc[] or DevSrcDesc[] (every message, generic kernels)
    std::unique_ptr<ohgpu::DevSrcDesc[], ohgpu::HostFree> host_descs;   // kBatchSrc: the same on the host (n of them); d_descs is made from it when the generic kernel first runs the whole batch
    mutable std::mutex lazy;      // ... under this
    const ohgpu_src* src;         // kBatchSrc only
    uint64_t src_arena_bytes, dst_arena_bytes;
    uint64_t in_frames, out_frames, src_bytes_touched, dst_bytes_written;
    uint32_t max_frames;          // largest n_frames in the batch
    bool     uniform;             // every descriptor has the same format fields
    bool     src_planar = false;  // resampled batches: the (uniform) source layout is OHGPU_FLAG_SRC_PLANAR32
    uint8_t  channels, src_bits, src_endian, dst_bits, dst_endian;
    ohgpu::SrcFastPlan fast;      // kBatchSrc only
    ohgpu::PcmLinePlan line;      // kBatchPcm only
    ohgpu::FlywheelPlan fly;      // kBatchFlywheel only
    ohgpu::FmtLinePlan fmtline;   // kBatchFmt only
    ohgpu::OhmPlan ohm;           // kBatchOhm only
    // kBatchSrc whose messages differ in layout: one uniform batch per layout (each with its own block-kernel plan), run one
    // after the other; this batch keeps every descriptor for the generic kernel (ohgpu_set_kernel_variant(1)).
    std::vector<ohgpu_batch*> parts;
    // A resampled batch's unit counters and a flywheel batch's workspace belong to ONE launch at a time.  Launches on the same
    // stream queue behind each other; a launch on another stream while the last one is still running is refused
    // (ohgpu_*_batch_run, OHGPU_ERR_INVALID) instead of silently sharing them.
    mutable hipStream_t last_stream = nullptr;
    mutable hipEvent_t  last_done = nullptr;
    // (the last launch carried the CALLER's events on its dispatch -- ohgpu_src_batch_run_timed -- and not last_done: "has it finished"
    // is then asked of last_stream itself)
    mutable bool last_untracked = false;
};

namespace ohgpu {

int set_error(int code, const char* fmt, ...);

#define OHGPU_HIP_TRY(expr)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return ::ohgpu::set_error(OHGPU_ERR_DEVICE, "%s failed: %s", #expr,  \
                                                        hipGetErrorString(e_));                     \
    } while (0)

// kernels
hipError_t launch_fmt_v1(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_pcm_v1(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_pcm_line(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
int plan_pcm_line(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_msg_desc* descs, size_t n,
                  const MsgPrefix* prefixes = nullptr, const uint8_t* blob = nullptr, size_t blob_bytes = 0);
// ohgpu_pcm_batch_create with a prefix per message (not part of the C ABI: the Songcast frame batch is its one user).
// (*out)->line.prefixed tells whether the line kernel took them; if not, nobody writes them.
int pcm_batch_create_prefixed(ohgpu_ctx* ctx, const ohgpu_msg_desc* descs, size_t n, uint64_t src_arena_bytes, uint64_t dst_arena_bytes,
                              const MsgPrefix* prefixes, const uint8_t* blob, size_t blob_bytes, ohgpu_batch** out);
void free_pcm_line(ohgpu_ctx* ctx, ohgpu_batch* b);
int plan_fmt_line(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_fmt_desc* descs, size_t n);
void free_fmt_line(ohgpu_ctx* ctx, ohgpu_batch* b);
hipError_t launch_fmt_line(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
// csrc/ohm_frame_kernel.hip: the two wire channels of streams wider than stereo (Sender::DoProcessFragment), one record per fragment
OhmSelRec wide_record(uint64_t src_off, uint64_t dst_off, uint32_t n_frames, uint32_t channels, uint32_t sb, bool little, uint64_t src_arena_bytes);
hipError_t launch_ohm_wide(const ohgpu_ctx* ctx, const void* d_recs, uint32_t n_recs, const uint8_t* src, uint8_t* dst, const uint8_t* prefix, hipStream_t s);
int plan_flywheel(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_flywheel_desc* descs, size_t n);
void free_flywheel(ohgpu_ctx* ctx, ohgpu_batch* b);
hipError_t ctx_dev_alloc(ohgpu_ctx* ctx, void** p, size_t bytes);     // csrc/ohgpu_api.hip: DevCache
void ctx_dev_free(ohgpu_ctx* ctx, void* p);
// The body of every ohgpu_*_process_host (csrc/ohgpu_api.hip): src_host goes to the context's source arena, `run` launches on the
// context's stream with the two device arenas, and the bytes the call's outputs cover -- `ranges` = (dst_offset, bytes) per output,
// any order -- come back: in one copy straight into dst_host when they tile a span of it, through the pinned bounce buffer run by
// run otherwise.  dst_host bytes no output covers are never written.  Synchronises.
int host_roundtrip(ohgpu_ctx* ctx, const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes,
                   std::vector<std::pair<uint64_t, uint64_t>>& ranges, const std::function<int(const void* d_src, void* d_dst)>& run);
hipError_t launch_flywheel(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_src_v1(const ohgpu_ctx* ctx, const void* d_descs, size_t n, const ohgpu_src* src_filter,
                         const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_src_block(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_src_lean(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_src_mfma(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s, uint32_t first_unit = 0);   // csrc/src_mfma_kernel.hip, legacy builds only (units [first_unit, n_lean))
// (`query`: nothing is launched; the instantiation the batch would run is asked what the device grants it -- ohgpu_src_batch_occupancy)
// (`start` / `stop`: the launch itself carries the two events -- hipExtLaunchKernelGGL: its dispatch's own timestamps, no packet more
// in the queue -- ohgpu_src_batch_run_timed)
struct WgOccupancy { bool query = true; int groups_per_cu = 0, designed_for = 0; uint32_t lds_bytes = 0; hipEvent_t start = nullptr, stop = nullptr; };
hipError_t launch_src_mfma_wg(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s, WgOccupancy* query = nullptr);   // csrc/src_mfma_wg_kernel.hip
bool src_mfma_wg_supported(uint32_t L_blk, uint32_t M_blk, uint32_t ch, uint32_t sb, uint32_t db, bool planar, bool halfband);
bool src_mfma_wg_unit_inside(int64_t src_row0, uint32_t row_src_bytes, uint64_t src_arena_bytes, uint32_t ch, uint32_t sb, uint32_t unit_rows, bool planar, uint64_t plane_stride, bool halfband);
bool build_mfma_halfband(const int32_t* coef_q28, uint32_t L_blk, std::vector<MfStep>* steps, std::vector<uint8_t>* amat);   // csrc/src_mfma_kernel.hip
bool build_mfma_tables(uint32_t L, uint32_t M, uint32_t T, const int32_t* coef_q28, uint32_t L_blk, uint32_t kb_cap,
                       std::vector<uint8_t>* adig, std::vector<MfStep>* steps);
void build_mfma_images(const std::vector<uint8_t>& adig, const std::vector<MfStep>& steps, uint32_t L, std::vector<uint8_t>* amat);
bool src_mfma_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t db);
void src_mfma_geometry(uint32_t* rows, uint32_t* wave_lds_bytes, uint32_t* max_waves);
uint32_t src_block_outputs(uint32_t L, uint32_t fb_dst);      // outputs per block (0: no block length fits): whole phase periods, >= 128, whole 64-byte lines
hipError_t launch_ramp_planes(const ohgpu_ctx* ctx, const void* d_jobs, uint32_t n_jobs, void* d_planes, hipStream_t s);   // csrc/ramp_plane_kernel.hip
hipError_t load_ramp_plane_kernel();
bool src_lean_geometry(uint32_t L, uint32_t T, bool halfband, uint32_t ch, uint32_t sb, uint32_t db, uint32_t out_per_drain,
                       uint32_t* rows, uint32_t* in_blocks, uint32_t* stage_frames, uint32_t* ring_bytes, uint32_t* coef_lds_bytes,
                       uint32_t* wave_lds_bytes, uint32_t* max_waves);
bool src_block_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le);
bool src_block_built(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le);   // ... and round 1's kernel itself is in this library for the layout (the fallback list; a legacy build: the whole list)
bool src_lean_only_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le);
bool src_lean_halfband_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le);
bool src_block_geometry(uint32_t L, uint32_t T, uint32_t ch, uint32_t sb, uint32_t db, uint32_t out_per_drain,
                        uint32_t* rows, uint32_t* ring_bytes, uint32_t* coef_lds_bytes, uint32_t* wave_lds_bytes, uint32_t* max_waves);

// host helpers
void build_ramp_table(uint16_t out[512]);
int  design_src(uint32_t rate_in, uint32_t rate_out, uint32_t T, double beta, double f_pass,
                std::vector<int32_t>* coef_q28, uint32_t* L, uint32_t* M);
// floor(t / d) for a divisor fixed over many t: a 64 x 64 -> 128 multiply by floor((2^64 - 1) / d) and at most two steps up (the
// estimate is never above and at most two below) -- a third of a hardware divide, and the planner's pass over half a million
// messages makes two a message.
struct FastDiv64 {
    uint64_t d, inv;
    explicit FastDiv64(uint64_t divisor) : d(divisor), inv(divisor > 1 ? ~0ull / divisor : 0) {}
    uint64_t div(uint64_t t) const
    {
        if (d <= 1) return t;
        uint64_t q = (uint64_t)(((unsigned __int128)t * inv) >> 64), r = t - q * d;
        while (r >= d) { q++; r -= d; }
        return q;
    }
};

// What a pass over messages [lo, hi) of a resampled batch finds (src_check_range, csrc/ohgpu_api.hip): the first bad descriptor's
// error, the batch's totals, whether the messages share descs[0]'s layout and come in the planner's order.
struct SrcRangeResult {
    int err = OHGPU_OK;
    char msg[512] = "";
    uint64_t in_frames = 0, out_frames = 0, src_bytes_touched = 0, dst_bytes_written = 0;
    uint32_t max_frames = 0;
    bool uniform = true;
    bool ordered = true;        // every message of the range is not before its predecessor in the planner's order (meaningful for a uniform batch)
    void fail(int code) { err = code; snprintf(msg, sizeof(msg), "%s", ohgpu_last_error()); }
};
void src_check_range(const ohgpu_src* src, const ohgpu_src_msg_desc* descs, size_t lo_i, size_t hi_i, uint64_t src_arena_bytes,
                     uint64_t dst_arena_bytes, DevSrcDesc* dev, SrcRangeResult* out);
// The usual message of a batch -- descs[0]'s layout (descs[0] has been through src_check_range), packed source, every test passed --
// checked in a few dozen cycles, in line, by the pass that also plans it: the eight bytes from `attenuation` to `flags` against
// descs[0]'s (all but the ramp and zero-LSB bits: what equals a validated message's is valid, and of its layout), the ranges, the
// window by FastDiv64; `r` gets the message's share of the totals, the caller its stream's two bases.  false = not that kind of
// message, nothing added: src_check_range says what it is (a bad one, one of another layout, or a good one of a rarer kind).
struct SrcQuickCheck {
    uint64_t L, M, T, src_arena, dst_arena, ok_word, fb_src, fb_dst;
    FastDiv64 by_L;
    bool usable;
    static constexpr uint64_t kWordMask = ~((uint64_t)(OHGPU_FLAG_RAMP | OHGPU_FLAG_ZERO_LSB32) << 56);
    static uint64_t word_of(const ohgpu_src_msg_desc& d) { uint64_t w; memcpy(&w, &d.attenuation, 8); return w; }
    SrcQuickCheck(uint64_t L_, uint64_t M_, uint64_t T_, const ohgpu_src_msg_desc& d0, uint64_t src_arena_bytes, uint64_t dst_arena_bytes)
        : L(L_), M(M_), T(T_), src_arena(src_arena_bytes), dst_arena(dst_arena_bytes), ok_word(word_of(d0) & kWordMask),
          fb_src((uint64_t)d0.channels * (d0.src_bits / 8)), fb_dst((uint64_t)d0.channels * (d0.dst_bits / 8)), by_L(L_),
          usable(!(d0.flags & OHGPU_FLAG_SRC_PLANAR32))
    {
        static_assert(offsetof(ohgpu_src_msg_desc, attenuation) == 48 && offsetof(ohgpu_src_msg_desc, flags) == 55 && sizeof(ohgpu_src_msg_desc) == 64, "the eight bytes from attenuation to flags");
    }
    __attribute__((always_inline)) bool pass(const ohgpu_src_msg_desc& d, SrcRangeResult& r, int64_t* sbase, int64_t* dbase) const
    {
        if ((word_of(d) & kWordMask) != ok_word || d.src_plane_stride != 0) return false;
        if (d.ramp_start > OHGPU_RAMP_MAX || d.ramp_end > OHGPU_RAMP_MAX || ((d.flags & OHGPU_FLAG_RAMP) && d.n_frames > 131071u)) return false;
        if (d.out_frame0 > (1ull << 48) || d.src_frame0 > (1ull << 48) || d.src_frames > (1ull << 40)) return false;
        const uint64_t src_bytes = d.src_frames * fb_src, dst_bytes = (uint64_t)d.n_frames * fb_dst;
        if (d.src_offset > src_arena || src_bytes > src_arena - d.src_offset || d.dst_offset > dst_arena || dst_bytes > dst_arena - d.dst_offset) return false;
        if (d.n_frames > 0) {
            const int64_t n0_first = (int64_t)by_L.div(d.out_frame0 * M), n0_last = (int64_t)by_L.div((d.out_frame0 + d.n_frames - 1) * M);
            const int64_t n_lo = n0_first - (int64_t)(T - 1);
            if (n_lo >= 0 ? (uint64_t)n_lo < d.src_frame0 : d.src_frame0 != 0) return false;
            if ((uint64_t)n0_last >= d.src_frame0 + d.src_frames) return false;
            r.in_frames += (uint64_t)(n0_last - n0_first + 1);
            r.src_bytes_touched += (uint64_t)(n0_last - (n_lo < 0 ? 0 : n_lo) + 1) * fb_src;
        }
        r.out_frames += d.n_frames;
        r.dst_bytes_written += dst_bytes;
        if (d.n_frames > r.max_frames) r.max_frames = d.n_frames;
        *sbase = (int64_t)d.src_offset - (int64_t)(d.src_frame0 * fb_src);
        *dbase = (int64_t)d.dst_offset - (int64_t)(d.out_frame0 * fb_dst);
        return true;
    }
};

// The planner checking the messages ITSELF, in the pass that cuts them into segments (a batch of half a million descriptors is 32 MB:
// a pass of its own over them is a third of the plan's time).  In: the filter (the arenas are the batch's).  Out: `checked` = every
// message was visited; `total` = what src_check_range found over all of them (its err / msg = the first bad descriptor's, in message
// order); `retry` = the messages are not what this pass assumes -- one layout, the planner's order -- and the caller must take the
// two-pass route (validation, then plan_src_fast with what it found).
struct PlanFusedCheck {
    const ohgpu_src* src = nullptr;
    bool checked = false, retry = false;
    SrcRangeResult total;
};
struct PlanDigest { uint64_t hash, units, pieces, ramp_jobs; int kernel; };   // ohgpu_src_plan_digest: a plan without a device
// `ordered`: the caller's messages are known to be in the planner's order already (src_msg_before never holds for a message against
// its predecessor: the validation pass looked), so the planner neither checks nor sorts
int  plan_src_fast(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_src_msg_desc* descs, size_t n, bool ordered, PlanDigest* digest = nullptr, PlanFusedCheck* fused = nullptr);

// The generic kernel's form of a (validated) resampled message: everything 64-bit that can be precomputed on the host.
inline DevSrcDesc src_convert_desc(const ohgpu_src_msg_desc& d, uint64_t L, uint64_t M)
{
    DevSrcDesc o;
    memset(&o, 0, sizeof(o));
    if (d.n_frames > 0) {
        const uint64_t t_first = d.out_frame0 * M;
        o.in_rel0 = (int64_t)(t_first / L) - (int64_t)d.src_frame0;
        o.phase0 = (uint32_t)(t_first % L);
    }
    o.src_offset = d.src_offset;
    o.dst_offset = d.dst_offset;
    o.n_frames = d.n_frames;
    o.ramp_i0 = 0;
    o.ramp_n = d.n_frames;
    o.ramp_start = d.ramp_start;
    o.ramp_end = d.ramp_end;
    o.channels = d.channels;
    o.src_bits = d.src_bits;
    o.src_endian = d.src_endian;
    o.dst_bits = d.dst_bits;
    o.dst_endian = d.dst_endian;
    o.flags = d.flags;
    o.plane_frames = (uint32_t)(d.src_plane_stride >> 2);
    return o;
}
// The planner's order of a uniform batch's messages: by stream -- identified by where its absolute frame 0 lives in the two arenas
// (and, planar, by the distance between its planes) -- then by output position.
inline bool src_msg_before(const ohgpu_src_msg_desc& x, const ohgpu_src_msg_desc& y, uint32_t fb_src, uint32_t fb_dst)
{
    const int64_t sx = (int64_t)x.src_offset - (int64_t)(x.src_frame0 * fb_src), sy = (int64_t)y.src_offset - (int64_t)(y.src_frame0 * fb_src);
    if (sx != sy) return sx < sy;
    if (x.src_plane_stride != y.src_plane_stride) return x.src_plane_stride < y.src_plane_stride;
    const int64_t dx = (int64_t)x.dst_offset - (int64_t)(x.out_frame0 * fb_dst), dy = (int64_t)y.dst_offset - (int64_t)(y.out_frame0 * fb_dst);
    if (dx != dy) return dx < dy;
    return x.out_frame0 < y.out_frame0;
}

void free_src_fast(ohgpu_ctx* ctx, ohgpu_batch* b);
// a batch's last launch: still running on a stream other than `s`?  /  wait for it
// (after a timed run nothing of the library's marks the launch's end, and the caller's stream may be gone by the time anyone asks:
// the device as a whole is waited for instead -- a diagnostic's price, paid only by a destroy, a rewrite of the ramps or a change of
// stream right behind a timed run)
inline bool batch_busy_on_another_stream(const ohgpu_batch* b, hipStream_t s)
{
    if (b->last_stream == s) return false;
    if (b->last_untracked) { (void)hipDeviceSynchronize(); b->last_untracked = false; return false; }
    return b->last_done != nullptr && hipEventQuery(b->last_done) == hipErrorNotReady;
}
inline hipError_t batch_wait_last_launch(const ohgpu_batch* b)
{
    if (b->last_untracked) { b->last_untracked = false; return hipDeviceSynchronize(); }
    return b->last_done ? hipEventSynchronize(b->last_done) : hipSuccess;
}
void free_ohm(ohgpu_ctx* ctx, ohgpu_batch* b);

}  // namespace ohgpu

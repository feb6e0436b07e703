// strk_api.hip — host side of libstrkit_amd.so (C ABI declared in include/strkit_amd.h).
// The batched counting path is here; three parts live in include fragments that are spliced into this file's
// anonymous namespace: strk_host_miss.inc (window-miss rounds), strk_host_ref.inc (reference side),
// strk_host_realign.inc (realignment).  strk_frontend.h holds the CPU-only record scan / read extraction.
//
// strk_dbam.inc (spliced in at the end) holds the device-side BGZF inflater.
//
// One context = one HIP device.  A batched call enqueues, on the caller's stream:
//   memset(counters) -> k_hash -> k_plan -> k_dp_band -> k_dp_band_wide -> k_dp_all -> k_dp_long -> k_dp_generic -> k_replay
//   -> counters D2H
// (strk_submit_loci_device) and is completed by strk_finish, which synchronises once.  Only when a read's search left its speculative candidate window (rare;
// strk_stats.n_miss_reads) does the host run extra rounds: re-score the wanted window on the
// device, replay that locus on the host with the same search_replay() the device uses.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/strkit_amd.h"
#include "strk_kernels.h"
#include "strk_realign.h"
#include "strk_frontend.h"
#include "strk_inflate.h"

extern "C" int strk_repeat_count(strk_ctx* ctx, int32_t start_count, const uint8_t* tr, int32_t tr_len, const uint8_t* fl,
                                 int32_t fl_len, const uint8_t* fr, int32_t fr_len, const uint8_t* motif, int32_t motif_len,
                                 int32_t max_iters, int32_t local_search_range, int32_t step_size, int32_t* out_cn,
                                 int32_t* out_score, int32_t* out_n_explored);

namespace {

thread_local std::string g_err;
constexpr int kWinStartLevel = 3, kWinLevels = 6;   // kWindowLevels below: a process starts at 8 sizes either side of the estimate
struct HostPipe;   // strk_host_pipe.inc: the pinned-slot pipeline behind strk_count_loci

// batched calls submitted and not yet finished, over all contexts of this process: a call that will share the
// device with others takes half the CU slots, so that the tail of one call and the head of the next co-run
std::atomic<int> g_calls_in_flight{0};

// Default candidate window of this process (one sample, whatever context a call runs on): level into kWindowLevels,
// and the number of consecutive default-window calls without a window miss since the level last changed.
// Misses cost extra rounds on the host: a call with more than a handful (> 0.4 % of its loci) moves a level up at
// once; eight (from the two widest levels: sixty-four) calls in a row with at most one miss per thousand loci move
// a level down — a probe, whose failures space the later ones out (g_win_failed).
// One level per motif-length bucket (win_bucket): the estimate round(|tr| / |motif|) is off by the read's indel drift divided
// by the motif length, so the reads of long motifs stay inside narrow windows that those of short ones leave.  The two
// narrowest levels (+-4, +-5) are open to the long-motif buckets only (kWinMinLevel).
std::atomic<int> g_win_level[strk::kWinBuckets] = {{kWinStartLevel}, {kWinStartLevel}, {kWinStartLevel}, {kWinStartLevel}, {kWinStartLevel}};
std::atomic<int> g_win_quiet[strk::kWinBuckets] = {{0}, {0}, {0}, {0}, {0}};
// A step down is a probe: the call after it either stays quiet or pays a window-miss round for every locus the narrower window
// does not hold (config 3, 3-4-base motifs at +-6: 111 loci, 18 ms on top of an 8 ms call) and steps up again.  Every failed
// probe doubles the number of quiet calls before the next one (64, 128, ... 4 096), per bucket; strk_adaptive_reset clears it.
std::atomic<int> g_win_probing[strk::kWinBuckets] = {{0}, {0}, {0}, {0}, {0}};   // 1: the level was last changed by a step down
std::atomic<int> g_win_failed[strk::kWinBuckets] = {{0}, {0}, {0}, {0}, {0}};    // probes that failed since the reset

// The band pass most recently enqueued by any context of this process (its kEvBand event): in whole-grid mode the next call's
// band pass waits for it (enqueue_scoring), so that two calls in flight run half a period apart whatever their submit times.
std::mutex g_band_chain_mu;
hipEvent_t g_band_chain_ev = nullptr;
int g_band_chain_dev = -1;

// CPUs this process may run on (a container's share, not the machine's core count)
int host_cpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) return CPU_COUNT(&set);
    return std::max(1u, std::thread::hardware_concurrency());
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? STRK_E_NOMEM : STRK_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // head_room: a quarter more than asked for, so that a buffer that grows call by call is not re-allocated every time;
    // the two multi-gigabyte buffers of an alignment file (strk_dbam.inc) take exactly what they need
    int ensure(size_t bytes, bool head_room = true) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = head_room ? bytes + bytes / 4 + 256 : bytes + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();   // the failed hipMalloc's error is sticky per thread: a retry with less memory must not meet it
            return fail(STRK_E_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        }
        cap = want;
        return 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct strk_ctx {
    int device = 0;
    // workspace
    DevBuf read_locus, win_lo, win_n, tab_off, table, cls_list, band_recs, band_recs_w, long_list, counters, scratch, state_i32, state_f64, spec, rhash, rep, exact;
    DevBuf win_lo2, win_n2, tab_off2, table2, items;
    DevBuf sc_dev;                  // strk_repeat_count's fast path: one read's arrays in one device buffer ...
    uint8_t* sc_host = nullptr;     // ... their pinned host image (one copy up) and the pinned result (one copy down)
    int4* sc_out = nullptr;
    // staging for the host-buffer entry points
    DevBuf in_seqs, in_seq_off, in_nfl, in_ntr, in_nfr, in_est, in_read_off, in_motifs, in_motif_off;
    DevBuf out_cn, out_score, out_n, out_start;
    // realignment (strk_realign)
    DevBuf rl_s1, rl_s2, rl_pairs, rl_trace, rl_edge, rl_out, rl_cigar, rl_queue;
    int32_t* h_counters = nullptr;  // pinned: counters + cells + scratch_used
    // a chain of events along one call: start | after k_hash + k_plan | after k_dp_band | after k_dp_band_wide | after the
    // first k_replay pass | after k_dp_all / k_dp_ref | after k_dp_long | after k_dp_generic | end (after k_replay and the
    // counters' copy)
    hipEvent_t ev[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_ints = 0;        // = kLongWaves * long_slot_ints + generic_ints
    size_t long_slot_ints = 0, generic_ints = 0;   // (set where the constants are known: strk_create)
    int band_cooldown = 0;   // > 0: the band kernel is switched off for that many calls (too many certificates failed)
    int band_penalty = 32;   // length of the next cool-down (doubles while retries keep failing)
    bool band_probation = true;   // the band has not proved itself on this context's data yet: only a sample of the reads takes it
    bool p_window_auto = false;
    int p_window_b[strk::kWinBuckets] = {0, 0, 0, 0, 0};   // the pending call's window per motif-length bucket (0: params.window for all)
    // work-queue lengths of the previous finished call (wave chunks), used to size the persistent grids of the
    // kernels that usually have little or nothing to do: an idle block still claims its 70-80 KB of LDS on a CU
    // and so delays the band blocks of the calls it overlaps with
    bool hist_valid = false;
    int hist_band_mode = 0, hist_reads = 1, hist_exact_chunks = 0, hist_wide_chunks = 0, hist_long = 0;
    bool hist_tail_heavy = false;   // the previous call had both band kernels busy (each more than a quarter of the other's cells)
    // one submitted-but-not-finished batched call (strk_submit_loci_device .. strk_finish)
    bool pending = false;
    strk_batch p_batch;
    strk_params p_params;
    strk_params p_params_in;       // as the caller gave them (a call that is run again goes through check_params again)
    int scratch_reruns = 0;        // the submitted call is a re-run of one that asked for more scratch than there was (grow_scratch)
    strk::KArgs p_args;
    strk::ReplayArgs p_replay;
    hipStream_t p_stream = nullptr;
    HostPipe* pipe = nullptr;   // created by the first large strk_count_loci call of this context
};

namespace {

using namespace strk;

constexpr int kDefaultWindow = 8;
enum { kEvStart = 0, kEvHead, kEvBand, kEvWide, kEvPre, kEvExact, kEvLong, kEvGeneric, kEvEnd, kNumEvents };
constexpr int kBandProbationReads = 2048;
// default half-widths of the candidate window, see g_win_level.  A search that converges at once scores start +- 4, so +-4 is
// the floor of the TABLE; tools/window_need.py (BASELINE config 4): motifs of 11+ bases never need more, 7-10 bases in 0.7 % of
// the loci, 5-6 bases in 6 %, 3-4 in 27 %.  Levels 4 and 5 exist for the long-motif buckets but are switched off (kWinMinLevel):
// measured in round 4 on config 4's shard (tools/cfg_probe.py, STRKIT_AMD_WINDOW_B), a narrower TABLE puts the band into a
// narrower class, and what that class lacks is the slack the certificate needs — +-6 everywhere 6.40 ms per call, +-5 for motifs
// of 7+ bases 6.34 ms (15 000 certificate failures per call instead of 500), +-4: 8.08 ms (62 000 failures).  The cells a narrow
// window saves are taken by laying the BAND around the table's inner candidates instead (strk_search.h: BandTune), which keeps
// the table's outer entries for the searches that the caller's feedback moves.
// The short-motif buckets stop at +-6: +-5 was tried there in round 3 (the search from a start the feedback moved by one size
// then ends at the window's edge, 780 reads per 10 000-locus call turn out uncertain: 205 M reads/s instead of 221 M).
constexpr int kWindowLevels[kWinLevels] = {4, 5, 6, 8, 11, 15};
// narrowest level a motif-length bucket may settle at.  Motifs of 1-2 bases stay at +-8: their estimate is off by a size for every
// second base of indel drift (tools/window_need.py, config 4: 0.3 % of those loci need more than +-6, none more than +-8), a miss is
// a host round of about a millisecond (config 4's shard: 7 missed reads, 2 ms of an 8.2 ms call), and the wider window costs such
// short motifs four more fork rows and no wider band class.
constexpr int kWinMinLevel[kWinBuckets] = {3, 2, 2, 2, 2};
// defaults of BandTune (strk_search.h): the band is laid around the table's middle +- kBandSpanW candidate sizes
constexpr int kBandSpanW = 64, kBandSlackM8 = 0;
// Scratch pool (int32 units): kLongWaves slots of kLongSlotInts for k_dp_long (one per resident wave; a slot
// holds the backward row of all column tiles + two boundary columns: windows up to ~16 kb), then 16 Mi
// ints of H rows for the generic kernel.  448 MiB of the 288 GB, allocated once per context.
constexpr int kBandBlocksPerCU = std::max(1, std::min(8, (160 * 1024) / (4 * kBandWaveLds + kLdsSlack + 1024)));
constexpr int kLongBlocks = 512, kLongWaves = kLongBlocks * 4;   // 2 waves per SIMD
constexpr size_t kLongSlotInts = (size_t)48 << 10;
constexpr size_t kGenericPoolInts = (size_t)16 << 20;
// what the two parts may grow to when a call asks for more (grow_scratch): a slot for the longest window classify() sends to
// k_dp_long (64 column tiles + two boundary columns of 2^20 rows: 18 GB for the 2 048 slots), 32 GiB of generic-kernel rows
constexpr size_t kLongSlotMaxInts = (size_t)kLongTile * kLongMaxTiles + 2 * (((size_t)1 << 20) + 256);
constexpr unsigned long long kGenericPoolMaxInts = 8ull << 30;
// device layout of the `counters` buffer: int32[kCntTotal] | pad | u64 cells | u64 scratch_used
constexpr size_t kCellsOff = 64 * sizeof(int32_t);
constexpr size_t kCountersBytes = kCellsOff + 10 * sizeof(unsigned long long);   // cells, scratch_used, band / exact / wide-band / long-kernel bytes, cells per kernel (kCell*)
// behind them, zeroed with them but not copied back: k_plan's census of the wide band classes and k_sort_wide's cursors
constexpr size_t kWideHistOff = (kCountersBytes + 63) & ~(size_t)63;
constexpr size_t kCountersAllBytes = kWideHistOff + 2 * kNumWideLists * 256 * sizeof(int32_t);

int check_params(const strk_params* p, strk_params* out) {
    if (!p) return fail(STRK_E_INVALID, "params is NULL");
    *out = *p;
    if (out->window <= 0) out->window = kDefaultWindow;
    if (2 * out->window + 1 > kTableMax) out->window = (kTableMax - 1) / 2;
    if (out->local_search_range < 0 || out->step_size < 1)
        return fail(STRK_E_INVALID, "local_search_range must be >= 0 and step_size >= 1");
    if (out->tie_rule != STRK_TIE_FIRST && out->tie_rule != STRK_TIE_LAST) return fail(STRK_E_INVALID, "bad tie_rule");
    if (out->end_flags < 0 || out->end_flags > 15) return fail(STRK_E_INVALID, "bad end_flags");
    static_assert(STRK_NARROW_NONE == strk::kNarrowNone && STRK_NARROW_DECREMENT == strk::kNarrowDecrement &&
                  STRK_NARROW_HALVE == strk::kNarrowHalve && STRK_NARROW_AFTER_SEED == strk::kNarrowAfterSeed, "include/strkit_amd.h <-> strk_search.h");
    if (out->narrowing < 0 || out->narrowing >= strk::kNarrowModes)
        return fail(STRK_E_INVALID, "narrowing schedule %d is not one of STRK_NARROW_NONE / _DECREMENT / _HALVE / _AFTER_SEED", out->narrowing);
    return 0;
}

// Common workspace sizing for a batch with n_reads / n_loci and at most n_items DP items.
int ensure_workspace(strk_ctx* c, int n_reads, int n_loci, size_t table_ints, size_t n_items) {
    const size_t nr = (size_t)std::max(n_reads, 1), nl = (size_t)std::max(n_loci, 1);
    int rc;
    if ((rc = c->read_locus.ensure(nr * 4))) return rc;
    if ((rc = c->win_lo.ensure(nr * 4))) return rc;
    if ((rc = c->win_n.ensure(nr * 4))) return rc;
    if ((rc = c->tab_off.ensure(nr * 8))) return rc;
    if ((rc = c->table.ensure(std::max<size_t>(table_ints, 1) * 4))) return rc;
    if ((rc = c->cls_list.ensure((size_t)kNumLists * std::max<size_t>(n_items, 1) * 2 * 4))) return rc;
    if ((rc = c->band_recs.ensure((size_t)kNumBandClasses * std::max<size_t>(n_items, 1) * 3 * sizeof(int4)))) return rc;
    if ((rc = c->counters.ensure(kCountersAllBytes))) return rc;
    if ((rc = c->band_recs_w.ensure((size_t)kNumBandClasses * std::max<size_t>(n_items, 1) * 3 * sizeof(int4)))) return rc;
    if ((rc = c->state_i32.ensure(nl * 4 * 4))) return rc;
    if ((rc = c->state_f64.ensure(nl * 8))) return rc;
    if ((rc = c->spec.ensure(nr * 16))) return rc;
    if ((rc = c->rhash.ensure(nr * 8))) return rc;
    if ((rc = c->rep.ensure(nr * 4))) return rc;
    if ((rc = c->exact.ensure(nr))) return rc;
    if (!c->scratch.p) {
        if ((rc = c->scratch.ensure(((size_t)kLongWaves * c->long_slot_ints + c->generic_ints) * 4))) return rc;
        c->scratch_ints = (size_t)kLongWaves * c->long_slot_ints + c->generic_ints;
    }
    return 0;
}

KArgs make_args(strk_ctx* c, const strk_batch* b, int end_flags, int window, int table_stride, int list_stride,
                const strk_params* sp) {
    KArgs a;
    memset(&a, 0, sizeof a);
    a.seqs = b->seqs; a.seq_off = b->seq_off; a.nfl = b->nfl; a.ntr = b->ntr; a.nfr = b->nfr;
    a.est_cn = b->est_cn; a.read_off = b->read_off; a.motifs = b->motifs; a.motif_off = b->motif_off;
    a.n_reads = b->n_reads; a.n_loci = b->n_loci;
    a.read_locus = c->read_locus.as<int32_t>();
    a.win_lo = c->win_lo.as<int32_t>();
    a.win_n = c->win_n.as<int32_t>();
    a.tab_off = c->tab_off.as<int64_t>();
    a.table = c->table.as<int32_t>();
    a.cls_list = c->cls_list.as<int32_t>();
    a.band_recs = c->band_recs.as<int4>();
    a.band_recs_w = a.band_recs;   // (k_sort_wide's copy when that kernel is launched: enqueue_scoring)
    a.wide_hist = reinterpret_cast<int32_t*>(c->counters.as<char>() + kWideHistOff);
    a.counters = c->counters.as<int32_t>();
    a.cells = reinterpret_cast<unsigned long long*>(c->counters.as<char>() + kCellsOff);
    a.scratch_used = a.cells + 1;
    a.scratch = c->scratch.as<int32_t>();
    a.scratch_cap = (long long)c->scratch_ints;
    a.long_slot = (long long)c->long_slot_ints;
    a.long_waves = kLongWaves;
    a.list_stride = list_stride;
    static const int dbg = getenv("STRKIT_AMD_DBG") ? atoi(getenv("STRKIT_AMD_DBG")) : 0;
    a.dbg = dbg;
    // tuning aids: STRKIT_AMD_SPAN_W / STRKIT_AMD_SLACK_M8 override where the forward band lies (strk_search.h: BandTune)
    static const int span_w = getenv("STRKIT_AMD_SPAN_W") ? atoi(getenv("STRKIT_AMD_SPAN_W")) : kBandSpanW;
    static const int slack_m8 = getenv("STRKIT_AMD_SLACK_M8") ? atoi(getenv("STRKIT_AMD_SLACK_M8")) : kBandSlackM8;
    a.band_tune.span_w = std::max(0, span_w);
    a.band_tune.slack_m8 = std::max(0, slack_m8);
    a.end_flags = end_flags;
    a.window = window;
    for (int k = 0; k < kWinBuckets; ++k) a.window_b[k] = c->p_window_b[k] > 0 ? c->p_window_b[k] : window;
    a.table_stride = table_stride;
    if (sp) {  // speculative search for start == est_cn inside the DP kernel
        a.spec = static_cast<int4*>(c->spec.p);
        a.max_iters = sp->max_iters; a.lsr = sp->local_search_range; a.step = sp->step_size;
        a.tie_last = sp->tie_rule == STRK_TIE_LAST;
        a.narrow = sp->narrowing;
        a.rep = c->rep.as<int32_t>();
        a.rhash = sp->no_dedupe ? nullptr : c->rhash.as<unsigned long long>();
        a.exact = c->exact.as<uint8_t>();
        a.band_mode = (!sp->no_band && c->band_cooldown == 0) ? 1 : 0;
        a.band_limit = c->band_probation ? kBandProbationReads : INT32_MAX;
    }
    return a;
}

// plan (classification) + all DP kernels for the reads in `items` (NULL = all reads).
void enqueue_scoring(strk_ctx* c, const KArgs& a, int mode, const int32_t* d_items, int n_items, int force_generic,
                     hipStream_t st, bool time_dp, const ReplayArgs* pre_replay = nullptr) {
    // Longest-first order of the wide band classes (k_sort_wide) pays when their chunks are few against the ~2 000 resident waves —
    // the launch then lasts as long as its longest chunk and whatever was started late (BASELINE config 5 at one GPU's share:
    // 6 400 chunks, k_dp_band_wide 3.77 -> 3.38 ms).  Two small kernels (strk_kernels.h: k_sort_wide_hist, k_sort_wide); run when
    // the previous call had at most 8 192 wide chunks (a first call sorts).  With more — config 4's shard: 16 000 — the order
    // still shortens k_dp_band_wide (2.03 -> 1.80 ms with ONE call on the device, sort included), but with two calls in flight the
    // two extra launches between the band kernels moved the calls into the pattern in which their band passes co-run: the
    // default bench gave 92.8 M reads/s twice where the unsorted queue gives 103-111 (profiles/r04_sort_wide_always_experiment.txt).
    const bool hist0 = mode == 0 && c->hist_valid && c->hist_band_mode == a.band_mode;
    const bool sort_wide = a.band_mode && mode == 0 && !force_generic && (!hist0 || (c->hist_wide_chunks > 0 && c->hist_wide_chunks <= 8192));
    hipLaunchKernelGGL(k_plan, dim3((n_items + 255) / 256), dim3(256), 0, st, a, mode, d_items, n_items, force_generic);
    static const int tune = getenv("STRKIT_AMD_DP_BLOCKS") ? atoi(getenv("STRKIT_AMD_DP_BLOCKS")) : 0;   // tuning aid
    // A call that shares the device with other calls in flight takes fifteen sixteenths of the CU slots per kernel: the free
    // slots are what lets the LDS-holding tail kernels of one call (k_dp_band_wide, k_dp_all, k_dp_long) start while another
    // call's band pass is resident; k_hash / k_plan / k_replay need no LDS and fit NEXT to two band waves per SIMD (2 x 184 of
    // 512 VGPRs).  tools/grid_sweep2.sh, two calls in flight: 448 blocks 239 M reads/s, 480 -> 246 M, 496 -> 240 M, 512 -> 186 M
    // (round 2, three calls in flight and 219 VGPRs: 7/8 was the best).
    // ... which pays when ONE kernel carries the call (BASELINE config 2: k_dp_band 1.15 ms, the others 0.13 ms; configs 3 and 5
    // alike: k_dp_all / k_dp_band_wide alone).  Where k_dp_band AND k_dp_band_wide are both large — config 4's shard: 3.4 and 2.0
    // ms — two calls' big kernels take turns on fifteen sixteenths of the chip each (measured with two calls in flight: 7.1 ms
    // per call against 6.55 ms with whole grids; the other way round for configs 3 and 5: 8.9 against 7.7 ms, 3.5 against 2.65 ms):
    // such a context takes the whole grid (hist_tail_heavy, from the previous call's cell counts).
    static const int force_heavy = getenv("STRKIT_AMD_FORCE_HEAVY") ? atoi(getenv("STRKIT_AMD_FORCE_HEAVY")) : -1;   // tuning aid: 1 / 0 pins the rule
    const bool heavy = force_heavy >= 0 ? force_heavy != 0 : (c->hist_valid && c->hist_tail_heavy);
    const int sixteenths = (g_calls_in_flight.load(std::memory_order_relaxed) > 1 && !heavy) ? 15 : 16;
    // tuning aid: STRKIT_AMD_GRID16="band,wide,exact" pins the sixteenths of the three persistent grids when calls overlap
    static const std::array<int, 3> pin16 = [] {
        std::array<int, 3> v{};
        if (const char* e = getenv("STRKIT_AMD_GRID16")) {
            for (int k = 0; k < 3 && *e; ++k) {
                v[k] = atoi(e);
                while (*e && *e != ',') ++e;
                if (*e == ',') ++e;
            }
        }
        return v;
    }();
    const bool overlap = g_calls_in_flight.load(std::memory_order_relaxed) > 1;
    // tuning aid: block slots a whole grid of an overlapping call leaves free (for the other call's small LDS-holding kernels)
    static const int spare_env = getenv("STRKIT_AMD_GRID_SPARE") ? atoi(getenv("STRKIT_AMD_GRID_SPARE")) : 0;
    const int spare = overlap ? std::max(0, spare_env) : 0;
    const int s16_band = (overlap && pin16[0] > 0) ? pin16[0] : sixteenths, s16_wide = (overlap && pin16[1] > 0) ? pin16[1] : sixteenths,
              s16_exact = (overlap && pin16[2] > 0) ? pin16[2] : sixteenths;
    // expected chunks of the sparsely used kernels, from the previous call of this context (same band mode), scaled
    // to this batch with 50 % head-room; without history every grid is the full resident one.  A grid that turns
    // out too small only makes that kernel slower: every wave pulls chunks until the queue is empty.
    const bool hist = mode == 0 && c->hist_valid && c->hist_band_mode == a.band_mode;
    auto predicted_blocks = [&](int chunks, int full) {
        if (!hist) return full;
        const double scaled = (double)chunks * std::max(1, a.n_reads) / std::max(1, c->hist_reads);
        const int blocks = chunks == 0 ? 1 : (int)(scaled * 1.5 / 4.0) + 2;
        return std::max(1, std::min(full, blocks));
    };
    if (time_dp) (void)hipEventRecord(c->ev[kEvHead], st);
    static const bool grid_dbg = getenv("STRKIT_AMD_GRID_DEBUG") != nullptr;
    if (grid_dbg && mode == 0)
        fprintf(stderr, "[strk grid] ctx %p reads %d in flight %d hist %d heavy %d -> sixteenths %d (wide chunks %d, exact chunks %d)\n", (void*)c, a.n_reads,
                g_calls_in_flight.load(), (int)c->hist_valid, (int)c->hist_tail_heavy, sixteenths, c->hist_wide_chunks, c->hist_exact_chunks);
    const bool band = a.band_mode && mode == 0 && !force_generic;
    int band_blocks = 1;
    if (band && time_dp) {
        // Whole-grid mode (two large kernels per call): which way two calls in flight share the device is decided by their phase.
        // Half a period apart, one call's k_dp_band_wide and tails run inside the other's k_dp_band (config 4's shard: 5.8 ms per
        // call); in phase, the two band passes co-run at half speed, then the two wide passes, and the tails of both are left with
        // nothing to hide in (6.9 ms).  Both patterns sustain themselves; two submits in a row on an idle device start the second
        // one.  So a band pass waits for the band pass enqueued before it, whichever context that was: the calls fall half a
        // period apart by themselves.
        std::lock_guard<std::mutex> lk(g_band_chain_mu);
        if (sixteenths == 16 && overlap && g_band_chain_ev && g_band_chain_ev != c->ev[kEvBand] && g_band_chain_dev == c->device)
            (void)hipStreamWaitEvent(st, g_band_chain_ev, 0);
    }
    if (band) {
        // banded first pass: certified reads are done, the others are appended to the exact lists below
        band_blocks = std::max(1, std::min(tune > 0 ? tune : std::max(1, 256 * kBandBlocksPerCU * s16_band / 16 - (s16_band == 16 ? spare : 0)), (a.list_stride + 3) / 4));
        hipLaunchKernelGGL(k_dp_band, dim3(band_blocks), dim3(256), 0, st, a);
    }
    if (time_dp) (void)hipEventRecord(c->ev[kEvBand], st);
    if (band && time_dp) {
        std::lock_guard<std::mutex> lk(g_band_chain_mu);
        g_band_chain_ev = c->ev[kEvBand];
        g_band_chain_dev = c->device;
    }
    if (band) {   // long windows
        KArgs aw = a;
        if (sort_wide) {
            hipLaunchKernelGGL(k_sort_wide_hist, dim3(kSortWideBlocks, kNumWideLists), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_sort_wide, dim3(kSortWideBlocks, kNumWideLists), dim3(256), 0, st, a, c->band_recs_w.as<int4>());
            aw.band_recs_w = c->band_recs_w.as<int4>();
        }
        const int wide_full = std::max(1, std::min(tune > 0 ? tune : std::max(1, 256 * kBandBlocksPerCU * s16_wide / 16 - (s16_wide == 16 ? spare : 0)), (a.list_stride + 3) / 4));
        hipLaunchKernelGGL(k_dp_band_wide, dim3(predicted_blocks(c->hist_wide_chunks, wide_full)), dim3(256), 0, st, aw);
    }
    if (time_dp) (void)hipEventRecord(c->ev[kEvWide], st);
    // first k_replay pass (strk_replay.h): as far as the certified band tables carry each locus, before the exact kernels
    if (band && pre_replay) hipLaunchKernelGGL(k_replay, dim3(a.n_loci), dim3(64), 0, st, a, *pre_replay);
    if (time_dp) (void)hipEventRecord(c->ev[kEvPre], st);
    if (!force_generic) {
        // persistent-style grid: every wave pulls chunks from the device-side queue until it is empty
        constexpr int kBlocksPerCU = std::max(1, std::min(8, (160 * 1024) / (4 * kWaveLdsBytes + kLdsSlack + 1024)));
        const int full = std::max(1, std::min(tune > 0 ? tune : 256 * kBlocksPerCU * s16_exact / 16, (a.list_stride + 3) / 4));
        const int blocks = a.ref_mode ? full : predicted_blocks(c->hist_exact_chunks, full);
        if (a.ref_mode) hipLaunchKernelGGL(k_dp_ref, dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_dp_all, dim3(blocks), dim3(256), 0, st, a);
    }
    if (time_dp) (void)hipEventRecord(c->ev[kEvExact], st);
    KArgs al = a;
    if (!force_generic && !a.ref_mode) {
        // longest first when there are more long items than resident waves (k_sort_long) — by the previous call's count, or
        // the item count the host knows (window-miss rounds, explicit tables); a call without them does not pay the launch
        const bool many_long = mode == 0 ? (!hist || c->hist_long > kLongWaves) : a.list_stride > kLongWaves;   // (no history: one block finds out)
        if (many_long && !c->long_list.ensure((size_t)std::max(1, a.list_stride) * 2 * 4)) {
            hipLaunchKernelGGL(k_sort_long, dim3(1), dim3(1024), 0, st, a, c->long_list.as<int32_t>());
            al.long_sorted = c->long_list.as<int32_t>();
        }
    }
    if (!force_generic && !a.ref_mode)   // (a window-miss round or an explicit table: never more blocks than items)
        hipLaunchKernelGGL(k_dp_long, dim3(mode == 0 ? predicted_blocks(c->hist_long, kLongBlocks) : std::max(1, std::min(kLongBlocks, a.list_stride))),
                           dim3(256), 0, st, al);
    if (time_dp) (void)hipEventRecord(c->ev[kEvLong], st);
    hipLaunchKernelGGL(k_dp_generic, dim3(1024), dim3(256), 0, st, a);   // one wave per (item, candidate): 4 096 waves
    if (time_dp) (void)hipEventRecord(c->ev[kEvGeneric], st);
}

// A call asked for more scratch than the context holds — rows of the generic kernel (its pool starts at 64 MiB), or a slot of
// k_dp_long for a window longer than ~20 000 candidate rows (a slot starts at 192 KiB): make that part as large as what was
// asked for (the device counted every request, served or not) and say that the call is to be run again.  Reads the generic
// kernel takes are rare (an empty flank, more than eight distinct symbols in a window, a motif longer than kMotifMax), and so
// are start counts dozens of times the tract's size, but a batch of a few hundred loci made of either must not fail for it.
bool grow_scratch(strk_ctx* c) {
    const int bits = c->h_counters[kCntError];
    if (!(bits & (kErrScratch | kErrLongSlot))) return false;
    size_t slot = c->long_slot_ints, gen = c->generic_ints;
    if (bits & kErrLongSlot) {
        const size_t need = (size_t)std::max(0, c->h_counters[kCntLongNeed]);
        if (need <= slot || need > kLongSlotMaxInts) return false;
        slot = (need + need / 8 + 1023) & ~(size_t)1023;
    }
    if (bits & kErrScratch) {
        const unsigned long long used = reinterpret_cast<const unsigned long long*>(reinterpret_cast<char*>(c->h_counters) + kCellsOff)[1];
        if (used <= gen || used > kGenericPoolMaxInts) return false;
        gen = (size_t)used + (1u << 16);
    }
    const size_t want = (size_t)kLongWaves * slot + gen;
    if (c->scratch.ensure(want * 4, false)) return false;
    c->long_slot_ints = slot;
    c->generic_ints = gen;
    c->scratch_ints = want;
    return true;
}

int check_error_bits(int bits) {
    if (bits & kErrBadInput) return fail(STRK_E_INVALID, "batch holds an empty motif or a negative length");
    if (bits & kErrScratch) return fail(STRK_E_NOMEM, "generic-kernel scratch exhausted (inputs too large for one call)");
    if (bits & kErrList) return fail(STRK_E_NOMEM, "a kernel's item list is full (more items than the call sized its lists for)");
    if (bits & kErrLongSlot) return fail(STRK_E_NOMEM, "a window is too long for the long-read kernel's scratch slot (|db| + 2 x candidate rows > %zu)", kLongSlotMaxInts);
    if (bits & kErrEmpty) return fail(STRK_E_EMPTY, "max() arg is an empty sequence: no candidate size could be scored for some read");
    return 0;
}

#include "strk_host_miss.inc"

// Enqueue one batched call on `st` and return without waiting.
int submit_device(strk_ctx* c, const strk_batch* b, const strk_params* params, int32_t* out_cn, int32_t* out_score,
                  int32_t* out_n, int32_t* out_start, hipStream_t st) {
    if (c->pending) return fail(STRK_E_INVALID, "context already holds a submitted call: strk_finish() it first");
    strk_params p;
    int rc;
    if ((rc = check_params(params, &p))) return rc;
    if (!b || b->n_reads < 0 || b->n_loci < 0) return fail(STRK_E_INVALID, "bad batch");
    // default window: 8 sizes either side of the estimate to begin with, then what the sample needs (g_win_level)
    c->p_window_auto = params->window <= 0;
    for (int k = 0; k < kWinBuckets; ++k) c->p_window_b[k] = 0;
    if (c->p_window_auto) {
        // tuning aid: STRKIT_AMD_WINDOW_B="w0,w1,w2,w3,w4" pins the default window of each motif-length bucket (0: adaptive)
        static const std::array<int, kWinBuckets> pinned = [] {
            std::array<int, kWinBuckets> v{};
            if (const char* e = getenv("STRKIT_AMD_WINDOW_B")) {
                for (int k = 0; k < kWinBuckets && *e; ++k) {
                    v[k] = atoi(e);
                    while (*e && *e != ',') ++e;
                    if (*e == ',') ++e;
                }
            }
            return v;
        }();
        p.window = 0;
        for (int k = 0; k < kWinBuckets; ++k) {
            const int w = pinned[k] > 0 ? std::min(pinned[k], kWindowLevels[kWinLevels - 1])
                                        : kWindowLevels[std::min(kWinLevels - 1, std::max(kWinMinLevel[k], g_win_level[k].load(std::memory_order_relaxed)))];
            c->p_window_b[k] = std::max(w, std::min(kWindowLevels[kWinLevels - 1], p.local_search_range + p.step_size));
            p.window = std::max(p.window, c->p_window_b[k]);
        }
    }
    c->p_batch = *b;
    c->p_params = p;
    c->p_params_in = *params;
    c->p_stream = st;
    if (b->n_reads == 0 || b->n_loci == 0) {
        c->pending = true;
        g_calls_in_flight.fetch_add(1, std::memory_order_relaxed);
        return 0;
    }
    if (!out_cn || !out_score || !out_n || !out_start) return fail(STRK_E_INVALID, "output pointer is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const int ts = std::min(kTableMax - 1, 2 * (p.window + 7) + 1);   // room for k_plan's per-read widening
    if ((rc = ensure_workspace(c, b->n_reads, b->n_loci, (size_t)b->n_reads * ts, (size_t)b->n_reads))) return rc;
    KArgs a = make_args(c, b, p.end_flags, p.window, ts, b->n_reads, &p);
    ReplayArgs rp;
    rp.max_iters = p.max_iters; rp.lsr = p.local_search_range; rp.step = p.step_size;
    rp.tie_last = p.tie_rule == STRK_TIE_LAST; rp.feedback = p.feedback; rp.narrow = p.narrowing;
    rp.out_cn = out_cn; rp.out_score = out_score; rp.out_n = out_n; rp.out_start = out_start;
    rp.next_read = c->state_i32.as<int32_t>();
    rp.need_lo = rp.next_read + b->n_loci;
    rp.need_hi = rp.need_lo + b->n_loci;
    rp.stop = rp.need_hi + b->n_loci;
    rp.frac = c->state_f64.as<double>();
    rp.resume = 0;
    rp.pre_exact = 0;

    struct InFlight {   // counted before the launches (the grid size depends on it), un-counted on any early error return
        bool keep = false;
        InFlight() { g_calls_in_flight.fetch_add(1, std::memory_order_relaxed); }
        ~InFlight() { if (!keep) g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed); }
    } in_flight;
    HIP_TRY(hipEventRecord(c->ev[kEvStart], st));
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, kCountersAllBytes, st));
    if (a.rhash) hipLaunchKernelGGL(k_hash, dim3((b->n_reads + 31) / 32), dim3(256), 0, st, a);   // eight lanes per read
    ReplayArgs rp_pre = rp;
    rp_pre.pre_exact = 1;
    enqueue_scoring(c, a, 0, nullptr, b->n_reads, 0, st, true, a.band_mode ? &rp_pre : nullptr);
    rp.resume = a.band_mode ? 1 : 0;   // band calls: the second pass, behind the exact kernels (strk_replay.h)
    hipLaunchKernelGGL(k_replay, dim3(b->n_loci), dim3(64), 0, st, a, rp);
    HIP_TRY(hipMemcpyAsync(c->h_counters, c->counters.p, kCountersBytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(c->ev[kEvEnd], st));
    HIP_TRY(hipGetLastError());
    c->p_args = a;
    c->p_replay = rp;
    c->pending = true;
    in_flight.keep = true;
    return 0;
}

// Wait for the submitted call, check for errors and resolve window misses.
int finish_device(strk_ctx* c, strk_stats* stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (!c->pending) return fail(STRK_E_INVALID, "nothing was submitted on this context");
    c->pending = false;
    g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed);
    const strk_batch* b = &c->p_batch;
    if (b->n_reads == 0 || b->n_loci == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[kEvEnd]));
    HIP_TRY(hipGetLastError());
    if (c->scratch_reruns < 3 && grow_scratch(c)) {   // the same call once more, with the scratch it asked for
        const strk_batch again = c->p_batch;
        const strk_params pin = c->p_params_in;
        const ReplayArgs rp = c->p_replay;
        ++c->scratch_reruns;
        int rc = submit_device(c, &again, &pin, rp.out_cn, rp.out_score, rp.out_n, rp.out_start, c->p_stream);
        if (!rc) rc = finish_device(c, stats);
        --c->scratch_reruns;
        return rc;
    }
    if (stats) {
        auto span = [&](int from, int to) {
            float ms = 0.f;
            return hipEventElapsedTime(&ms, c->ev[from], c->ev[to]) == hipSuccess ? ms : 0.f;
        };
        stats->kernel_ms = span(kEvStart, kEvEnd);
        stats->head_ms = span(kEvStart, kEvHead);
        stats->band_kernel_ms = span(kEvHead, kEvBand);
        stats->band_wide_kernel_ms = span(kEvBand, kEvWide);
        stats->dp_kernel_ms = span(kEvPre, kEvExact);
        stats->long_kernel_ms = span(kEvExact, kEvLong);
        stats->generic_kernel_ms = span(kEvLong, kEvGeneric);
        stats->replay_ms = span(kEvGeneric, kEvEnd) + span(kEvWide, kEvPre);   // both k_replay passes
        {
            const unsigned long long* u = reinterpret_cast<const unsigned long long*>(reinterpret_cast<char*>(c->h_counters) + kCellsOff);
            stats->band_bytes = (int64_t)u[2];
            stats->exact_bytes = (int64_t)u[3];
            stats->wide_bytes = (int64_t)u[4];
            stats->long_bytes = (int64_t)u[5];
            stats->band_cells = (int64_t)u[kCellBand];
            stats->wide_cells = (int64_t)u[kCellWide];
            stats->exact_cells = (int64_t)u[kCellExact];
            stats->long_cells = (int64_t)u[kCellLong];
        }
        stats->n_long_reads = c->h_counters[kCntClass0 + kLongClass];
        stats->n_dp_launches = 2;
        stats->window_used = c->p_params.window;
        for (int k = 0; k < kWinBuckets; ++k)   // the window each motif-length bucket ran with (0: no locus of that bucket in the call)
            stats->window_bucket[k] = c->h_counters[kCntLociB + k] > 0 ? (c->p_window_b[k] > 0 ? c->p_window_b[k] : c->p_params.window) : 0;
        if (c->p_window_auto) {   // the widest default window among the motif-length buckets that had loci in this call
            int w = 0;
            for (int k = 0; k < kWinBuckets; ++k) w = std::max(w, stats->window_bucket[k]);
            if (w > 0) stats->window_used = w;
        }
        stats->n_fallback = c->h_counters[kCntClass0 + kGenericClass];
        stats->n_dedup_reads = c->h_counters[kCntDup];
        stats->n_band_reads = 0;
        for (int k = 0; k < kNumBandClasses; ++k) stats->n_band_reads += c->h_counters[kCntClass0 + kBandClass0 + k];
        stats->n_band_fallback = c->h_counters[kCntBandFallback];
        stats->dp_cells = (int64_t) * reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->h_counters) + kCellsOff);
    }
#ifdef STRK_PHASE_TIMING
    fprintf(stderr, "[phase ticks/64] header %d stage %d (of which pads %d, window bytes %d) tables %d bwd %d fwd %d epilogue %d\n", c->h_counters[48],
            c->h_counters[49] + c->h_counters[54] + c->h_counters[55], c->h_counters[55], c->h_counters[54], c->h_counters[50], c->h_counters[51],
            c->h_counters[52], c->h_counters[53]);
    fprintf(stderr, "[phase] longest chunk %d ticks/64 (%.3f ms at 2.4 GHz), most rows in a chunk %d, chunks %d\n", c->h_counters[45],
            c->h_counters[45] * 64.0 / 2.4e6, c->h_counters[46], c->h_counters[47]);
#endif
    int n_band_reads = 0;
    for (int k = 0; k < kNumBandClasses; ++k) n_band_reads += c->h_counters[kCntClass0 + kBandClass0 + k];
    // (a call whose band certificates mostly failed reports those reads as misses too: not a window problem)
    const bool band_unhealthy = c->p_args.band_mode && n_band_reads >= 64 && 2 * c->h_counters[kCntBandFallback] > n_band_reads;
    if (c->p_window_auto && !band_unhealthy) {
        for (int k = 0; k < kWinBuckets; ++k) {
            const int n_loci_k = c->h_counters[kCntLociB + k], n_miss = c->h_counters[kCntMissB + k];
            if (n_loci_k == 0) continue;
            const int level = std::max(kWinMinLevel[k], g_win_level[k].load(std::memory_order_relaxed));
            // a handful of misses costs less (one short extra round) than a wider window for every read does
            // (small calls: two loci of 250 already are 0.8 %, and a window-miss round on long windows costs as much as the call)
            const int failed = g_win_failed[k].load(std::memory_order_relaxed);
            const int quiet_calls = failed > 0 ? 64 << std::min(failed - 1, 6) : (level > kWinStartLevel ? 64 : 8);
            if (n_miss > std::max(1, n_loci_k / 250)) {
                if (level < kWinLevels - 1) g_win_level[k].store(level + 1, std::memory_order_relaxed);
                if (g_win_probing[k].exchange(0, std::memory_order_relaxed)) g_win_failed[k].store(std::min(failed + 1, 16), std::memory_order_relaxed);
                g_win_quiet[k].store(0, std::memory_order_relaxed);
            } else if (n_miss > n_loci_k / 1000) {   // more than one locus in a thousand: not a quiet call
                g_win_quiet[k].store(0, std::memory_order_relaxed);
            } else if (g_win_quiet[k].fetch_add(1, std::memory_order_relaxed) + 1 >= quiet_calls && level > kWinMinLevel[k]) {
                g_win_level[k].store(level - 1, std::memory_order_relaxed);
                g_win_quiet[k].store(0, std::memory_order_relaxed);
                g_win_probing[k].store(1, std::memory_order_relaxed);
            }
        }
    }
    {   // queue lengths of this call, for the grids of the next one (enqueue_scoring)
        int exact_chunks = 0, wide_chunks = 0;
        for (int k = 0; k < kNumClasses; ++k) {
            const int per = 64 / class_G(k);
            exact_chunks += (c->h_counters[kCntClass0 + k] + per - 1) / per;
        }
        for (int k = 0; k < kNumBandClasses; ++k) {   // the classes of k_dp_band_wide
            if (!band_class_wide_kernel(k)) continue;
            const int per = 64 / band_class_G(k);
            wide_chunks += (c->h_counters[kCntClass0 + kBandClass0 + k] + per - 1) / per;
        }
        // (a call on band probation sent all but its first reads to the exact kernels: its queue lengths say nothing about the
        // next call's — a one-block grid then crawled through 30 000 wide-band chunks in 46 ms, profiles/README.md round 3)
        c->hist_valid = !(c->p_args.band_mode && c->p_args.band_limit != INT32_MAX);
        c->hist_band_mode = c->p_args.band_mode;
        c->hist_reads = std::max(1, b->n_reads);
        c->hist_exact_chunks = exact_chunks;
        c->hist_wide_chunks = wide_chunks;
        c->hist_long = c->h_counters[kCntClass0 + kLongClass];
        {   // (cells, not event spans: with calls in flight a span includes the wait for the other call's kernels)
            const unsigned long long* u = reinterpret_cast<const unsigned long long*>(reinterpret_cast<char*>(c->h_counters) + kCellsOff);
            c->hist_tail_heavy = u[kCellWide] > u[kCellBand] / 4 && u[kCellBand] > u[kCellWide] / 4;
        }
    }
    {   // adaptive: noisy reads mostly fail the certificate and pay for both passes.  A context starts on probation
        // (the band sees the first kBandProbationReads reads of a call only, so a failure is cheap); a call with fewer
        // than half of its band reads falling back ends it, one with more switches the band off for a while and for
        // twice as long every time a retry (again on probation) fails.
        int nb = 0;
        for (int k = 0; k < kNumBandClasses; ++k) nb += c->h_counters[kCntClass0 + kBandClass0 + k];
        static const bool aid = getenv("STRKIT_AMD_DBG") && atoi(getenv("STRKIT_AMD_DBG")) != 0;
        if (aid) {   // profiling runs with parts of the kernels switched off (wrong scores by design): keep the band on
            c->band_cooldown = 0;
            c->band_probation = false;
        } else if (c->band_cooldown > 0) {
            --c->band_cooldown;
        } else if (nb >= 64) {
            if (2 * c->h_counters[kCntBandFallback] > nb) {
                c->band_cooldown = c->band_penalty;
                c->band_penalty = std::min(c->band_penalty * 2, 1 << 14);
                c->band_probation = true;
            } else {
                c->band_penalty = 32;
                c->band_probation = false;
            }
        }
    }
    const int err = c->h_counters[kCntError];
    int rc;
    if ((rc = check_error_bits(err & ~kErrEmpty))) return rc;
    if (c->h_counters[kCntMiss] > 0) return resolve_misses(c, b, c->p_params, c->p_args, c->p_replay, c->p_stream, stats, err);
    return check_error_bits(err);
}

int count_device(strk_ctx* c, const strk_batch* b, const strk_params* params, int32_t* out_cn, int32_t* out_score,
                 int32_t* out_n, int32_t* out_start, hipStream_t st, strk_stats* stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    const int rc = submit_device(c, b, params, out_cn, out_score, out_n, out_start, st);
    if (rc) return rc;
    return finish_device(c, stats);
}

// uploads a host batch into the context's staging buffers; returns a batch of device pointers
// `d_seqs` (optional): the bases are in device memory already (strk_dbam_extract); b->seqs is not read then
int upload_batch(strk_ctx* c, const strk_batch* b, strk_batch* d, hipStream_t st, const uint8_t* d_seqs = nullptr) {
    if (!b || b->n_reads < 0 || b->n_loci < 0) return fail(STRK_E_INVALID, "bad batch");
    *d = *b;
    if (b->n_reads == 0 || b->n_loci == 0) return 0;
    if (!b->seq_off || !b->nfl || !b->ntr || !b->nfr || !b->read_off || !b->motifs || !b->motif_off)
        return fail(STRK_E_INVALID, "batch pointer is NULL");
    const size_t nr = (size_t)b->n_reads, nl = (size_t)b->n_loci;
    if (b->read_off[0] != 0 || b->read_off[nl] != b->n_reads) return fail(STRK_E_INVALID, "read_off must span [0, n_reads]");
    for (size_t l = 0; l < nl; ++l) {
        if (b->read_off[l + 1] < b->read_off[l]) return fail(STRK_E_INVALID, "read_off must be non-decreasing");
        if (b->motif_off[l + 1] <= b->motif_off[l]) return fail(STRK_E_INVALID, "locus %zu has an empty motif", l);
    }
    for (size_t r = 0; r < nr; ++r) {
        if (b->nfl[r] < 0 || b->ntr[r] < 0 || b->nfr[r] < 0) return fail(STRK_E_INVALID, "read %zu has a negative length", r);
        if (b->seq_off[r + 1] - b->seq_off[r] != (int64_t)b->nfl[r] + b->ntr[r] + b->nfr[r])
            return fail(STRK_E_INVALID, "read %zu: seq_off does not match nfl+ntr+nfr", r);
    }
    const size_t nbases = (size_t)b->seq_off[nr], nmot = (size_t)b->motif_off[nl];
    if (nbases && !b->seqs && !d_seqs) return fail(STRK_E_INVALID, "seqs is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
#define UP(buf, src, bytes, field)                                                              \
    if ((rc = c->buf.ensure(std::max<size_t>((bytes), 16)))) return rc;                         \
    if ((bytes) > 0) HIP_TRY(hipMemcpyAsync(c->buf.p, (src), (bytes), hipMemcpyHostToDevice, st)); \
    d->field = static_cast<decltype(d->field)>(c->buf.p);
    if (d_seqs) d->seqs = d_seqs;
    else { UP(in_seqs, b->seqs, nbases, seqs) }
    UP(in_seq_off, b->seq_off, (nr + 1) * 8, seq_off)
    UP(in_nfl, b->nfl, nr * 4, nfl)
    UP(in_ntr, b->ntr, nr * 4, ntr)
    UP(in_nfr, b->nfr, nr * 4, nfr)
    UP(in_read_off, b->read_off, (nl + 1) * 4, read_off)
    UP(in_motifs, b->motifs, nmot, motifs)
    UP(in_motif_off, b->motif_off, (nl + 1) * 4, motif_off)
    if (b->est_cn) {
        UP(in_est, b->est_cn, nr * 4, est_cn)
    } else {
        if ((rc = c->in_est.ensure(nr * 4))) return rc;
        HIP_TRY(hipMemsetAsync(c->in_est.p, 0, nr * 4, st));
        d->est_cn = c->in_est.as<int32_t>();
    }
#undef UP
    return 0;
}


#include "strk_host_pipe.inc"

// strk_repeat_count's fast path: ONE copy up (the read's arrays in one pinned image), k_scalar_plan + k_dp_all on one block,
// ONE copy down (the search result) — instead of the batched path's nine uploads, dozen launches and four downloads for a
// single read (about 0.1 ms).  Returns 0 with the result, 1 when the read has to go the general way (no fast class, more than
// eight symbol classes, a search that leaves the window, nothing scored), < 0 on an error.
constexpr size_t kScalarSeqMax = 1792, kScalarMotifMax = 256;
constexpr size_t kScOffSeqOff = 0, kScOffLens = 16, kScOffReadOff = 32, kScOffMotifOff = 40, kScOffMotif = 48,
                 kScOffSeq = kScOffMotif + kScalarMotifMax, kScBytes = kScOffSeq + kScalarSeqMax + 64;
int scalar_fast(strk_ctx* c, int32_t start, const uint8_t* tr, int32_t ntr, const uint8_t* fl, int32_t nfl, const uint8_t* fr, int32_t nfr,
                const uint8_t* motif, int32_t m, int32_t max_iters, int32_t lsr, int32_t step, int32_t window, int32_t* cn, int32_t* score,
                int32_t* n_explored) {
    static const bool off = getenv("STRKIT_AMD_NO_SCALAR_FAST") != nullptr;
    const size_t ndb = (size_t)nfl + ntr + nfr;
    if (off || c->pending || nfl < 1 || nfr < 1 || ndb + 1 > kScalarSeqMax || (size_t)m > kScalarMotifMax || lsr < 0 || step < 1) return 1;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if (!c->sc_host) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->sc_host), kScBytes, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->sc_out), 64, hipHostMallocDefault));
    }
    if ((rc = c->sc_dev.ensure(kScBytes))) return rc;
    const int ts = std::min(kTableMax - 1, 2 * (window + 7) + 1);
    if ((rc = ensure_workspace(c, 1, 1, (size_t)ts, 1))) return rc;
    uint8_t* h = c->sc_host;
    const int64_t seq_off[2] = {0, (int64_t)ndb};
    const int32_t lens[4] = {nfl, ntr, nfr, start}, read_off[2] = {0, 1}, motif_off[2] = {0, m};
    memcpy(h + kScOffSeqOff, seq_off, 16);
    memcpy(h + kScOffLens, lens, 16);
    memcpy(h + kScOffReadOff, read_off, 8);
    memcpy(h + kScOffMotifOff, motif_off, 8);
    memcpy(h + kScOffMotif, motif, (size_t)m);
    memcpy(h + kScOffSeq, fl, (size_t)nfl);
    if (ntr) memcpy(h + kScOffSeq + nfl, tr, (size_t)ntr);
    memcpy(h + kScOffSeq + nfl + ntr, fr, (size_t)nfr);
    hipStream_t st = nullptr;
    const size_t up = kScOffSeq + ndb;
    HIP_TRY(hipMemcpyAsync(c->sc_dev.p, h, up, hipMemcpyHostToDevice, st));
    const uint8_t* d = c->sc_dev.as<uint8_t>();
    strk_batch b;
    b.n_reads = 1; b.n_loci = 1;
    b.seqs = d + kScOffSeq; b.seq_off = reinterpret_cast<const int64_t*>(d + kScOffSeqOff);
    b.nfl = reinterpret_cast<const int32_t*>(d + kScOffLens); b.ntr = b.nfl + 1; b.nfr = b.nfl + 2; b.est_cn = b.nfl + 3;
    b.read_off = reinterpret_cast<const int32_t*>(d + kScOffReadOff);
    b.motifs = d + kScOffMotif; b.motif_off = reinterpret_cast<const int32_t*>(d + kScOffMotifOff);
    strk_params p;
    memset(&p, 0, sizeof p);
    p.max_iters = max_iters; p.local_search_range = lsr; p.step_size = step; p.tie_rule = STRK_TIE_FIRST; p.end_flags = STRK_SG_ALL;
    p.no_dedupe = 1; p.no_band = 1;
    for (int k = 0; k < kWinBuckets; ++k) c->p_window_b[k] = 0;
    KArgs a = make_args(c, &b, p.end_flags, window, ts, 1, &p);
    hipLaunchKernelGGL(k_scalar_plan, dim3(1), dim3(64), 0, st, a, (int)(kCountersBytes / 4));
    hipLaunchKernelGGL(k_dp_all, dim3(1), dim3(256), 0, st, a);
    HIP_TRY(hipMemcpyAsync(c->sc_out, a.spec, sizeof(int4), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    const int4 r = *c->sc_out;
    if (r.w != 0) return 1;   // miss / nothing scored / not scored at all: the general path decides (and reports)
    *cn = r.x; *score = r.y; *n_explored = r.z;
    return 0;
}

// strk_score_table / strk_score_ref_table: explicit candidate windows per read, HOST buffers.
// ref_mode = 1 scores the reference-side candidate fl + motif*i (no right flank) and also returns
// the db position where the alignment ends (repeats.py:23-43).
int score_table_impl(strk_ctx* ctx, const strk_batch* batch, const int32_t* lo, const int32_t* n,
                     const int64_t* table_off, int32_t end_flags, int32_t force_generic, int32_t ref_mode,
                     int32_t* scores, int32_t* end_query, strk_stats* stats) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    if (ctx->pending) return fail(STRK_E_INVALID, "context holds a submitted call: strk_finish() it first");
    if (stats) memset(stats, 0, sizeof *stats);
    if (end_flags < 0 || end_flags > 15) return fail(STRK_E_INVALID, "bad end_flags");
    strk_batch d;
    int rc;
    if ((rc = upload_batch(ctx, batch, &d, nullptr))) return rc;
    if (batch->n_reads == 0 || batch->n_loci == 0) return 0;
    if (!lo || !n || !table_off || !scores) return fail(STRK_E_INVALID, "lo / n / table_off / scores is NULL");
    const size_t nr = (size_t)batch->n_reads;
    const int mul = ref_mode ? 2 : 1;
    size_t n_chunks = 0;
    for (size_t r = 0; r < nr; ++r) {
        if (lo[r] < 0 || n[r] < 0) return fail(STRK_E_INVALID, "read %zu: negative window", r);
        if (table_off[r + 1] - table_off[r] < n[r] || table_off[r] < 0) return fail(STRK_E_INVALID, "read %zu: table_off too small", r);
        n_chunks += ((size_t)n[r] + kTableMax - 1) / kTableMax;
    }
    const size_t tab = (size_t)table_off[nr];
    if ((rc = ensure_workspace(ctx, batch->n_reads, batch->n_loci, tab * mul, std::max<size_t>(n_chunks, 1)))) return rc;
    KArgs a = make_args(ctx, &d, end_flags, 0, 0, (int)std::max<size_t>(n_chunks, 1), nullptr);
    a.ref_mode = ref_mode;
    hipStream_t st = nullptr;
    std::vector<int64_t> off_dev(table_off, table_off + nr);
    for (auto& o : off_dev) o *= mul;
    HIP_TRY(hipMemcpyAsync(a.win_lo, lo, nr * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(a.win_n, n, nr * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(a.tab_off, off_dev.data(), nr * 8, hipMemcpyHostToDevice, st));
    for (int attempt = 0;; ++attempt) {
        HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, kCountersBytes, st));
        HIP_TRY(hipEventRecord(ctx->ev[kEvStart], st));
        enqueue_scoring(ctx, a, 1, nullptr, batch->n_reads, force_generic, st, true);
        HIP_TRY(hipEventRecord(ctx->ev[kEvEnd], st));
        HIP_TRY(hipMemcpyAsync(ctx->h_counters, ctx->counters.p, kCountersBytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        if (attempt < 3 && grow_scratch(ctx)) {   // once more, with the scratch the call asked for
            a.scratch = ctx->scratch.as<int32_t>();
            a.scratch_cap = (long long)ctx->scratch_ints;
            a.long_slot = (long long)ctx->long_slot_ints;
            continue;
        }
        break;
    }
    if ((rc = check_error_bits(ctx->h_counters[kCntError]))) return rc;
    if (tab) {
        if (!ref_mode) {
            HIP_TRY(hipMemcpy(scores, a.table, tab * 4, hipMemcpyDeviceToHost));
        } else {
            std::vector<int32_t> pairs(tab * 2);
            HIP_TRY(hipMemcpy(pairs.data(), a.table, tab * 8, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < tab; ++k) {
                scores[k] = pairs[2 * k];
                end_query[k] = pairs[2 * k + 1];
            }
        }
    }
    if (stats) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[kEvStart], ctx->ev[kEvEnd]) == hipSuccess) stats->kernel_ms = ms;
        if (hipEventElapsedTime(&ms, ctx->ev[kEvPre], ctx->ev[kEvExact]) == hipSuccess) stats->dp_kernel_ms = ms;
        stats->n_dp_launches = 2;
        stats->n_fallback = ctx->h_counters[kCntClass0 + kGenericClass];
        stats->dp_cells = (int64_t) * reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->h_counters) + kCellsOff);
    }
    return 0;
}

// Lazily scored (fwd score, fwd end_query, rev score, rev end_query) per candidate size for one locus:
// two device "reads" — the window itself and its reversal with the flanks swapped
// (repeats.py:32-41: ext_l_seq = (tr_candidate + flank_right_seq)[::-1] against db_seq[::-1]).
#include "strk_host_ref.inc"

#include "strk_host_realign.inc"

}  // namespace

extern "C" {

const char* strk_last_error(void) { return g_err.c_str(); }
const char* strk_version(void) { return "strkit_amd 0.1.0 (gfx950)"; }

void strk_adaptive_reset(void) {
    for (int k = 0; k < strk::kWinBuckets; ++k) {
        g_win_level[k].store(kWinStartLevel, std::memory_order_relaxed);
        g_win_quiet[k].store(0, std::memory_order_relaxed);
        g_win_probing[k].store(0, std::memory_order_relaxed);
        g_win_failed[k].store(0, std::memory_order_relaxed);
    }
}

int strk_host_register(void* ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return fail(STRK_E_INVALID, "strk_host_register: null pointer or no bytes");
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return 0; }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(STRK_E_DEVICE, "hipHostRegister(%lld bytes): %s", (long long)bytes, hipGetErrorString(e)); }
    return 0;
}

int strk_host_is_pinned(const void* ptr, int64_t bytes) { return bytes > 0 && host_range_pinned(ptr, (size_t)bytes) ? 1 : 0; }

int strk_host_unregister(void* ptr) {
    if (!ptr) return fail(STRK_E_INVALID, "strk_host_unregister: null pointer");
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(STRK_E_DEVICE, "hipHostUnregister: %s", hipGetErrorString(e)); }
    return 0;
}

int strk_device_mem(int device, int64_t* free_bytes, int64_t* total_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(STRK_E_NODEV, "no HIP device visible");
    if (device < 0 || device >= n) return fail(STRK_E_NODEV, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}

int strk_init(int device, strk_ctx** out) {
    if (!out) return fail(STRK_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(STRK_E_NODEV, "no HIP device visible");
    if (device < 0 || device >= n) return fail(STRK_E_NODEV, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(STRK_E_NODEV, "device %d is %s; this library holds gfx950 code objects only", device, prop.gcnArchName);
    strk_ctx* c = new strk_ctx();
    c->device = device;
    c->long_slot_ints = kLongSlotInts;
    c->generic_ints = kGenericPoolInts;
    // testing aids: a context that starts with small scratch parts, so that a small input walks the grow-and-run-again path
    if (const char* e = getenv("STRKIT_AMD_GENERIC_POOL_INTS")) c->generic_ints = (size_t)std::max(1l, atol(e));
    if (const char* e = getenv("STRKIT_AMD_LONG_SLOT_INTS")) c->long_slot_ints = (size_t)std::max(1l, atol(e));
    strk::ScoreTables t;
    strk::build_score_tables(&t);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(strk::c_mat), t.mat, sizeof t.mat);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(strk::c_enc), t.enc, sizeof t.enc);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_counters), kCountersBytes, hipHostMallocDefault);
    for (int i = 0; i < kNumEvents && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        strk_destroy(c);
        return fail(STRK_E_DEVICE, "context setup: %s", hipGetErrorString(e));
    }
    *out = c;
    return 0;
}

void strk_destroy(strk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    pipe_destroy(c->pipe);
    c->pipe = nullptr;
    DevBuf* bufs[] = {&c->read_locus, &c->win_lo, &c->win_n, &c->tab_off, &c->table, &c->cls_list, &c->band_recs, &c->counters,
                      &c->scratch, &c->state_i32, &c->state_f64, &c->spec, &c->rhash, &c->rep, &c->exact, &c->win_lo2, &c->win_n2, &c->tab_off2, &c->table2,

                      &c->items, &c->in_seqs, &c->in_seq_off, &c->in_nfl, &c->in_ntr, &c->in_nfr, &c->in_est,
                      &c->in_read_off, &c->in_motifs, &c->in_motif_off, &c->out_cn, &c->out_score, &c->out_n,
                      &c->out_start, &c->rl_s1, &c->rl_s2, &c->rl_pairs, &c->rl_trace, &c->rl_edge, &c->rl_out, &c->rl_cigar,
                      &c->rl_queue, &c->band_recs_w, &c->sc_dev, &c->long_list};
    for (DevBuf* b : bufs) b->release();
    {
        std::lock_guard<std::mutex> lk(g_band_chain_mu);
        if (g_band_chain_ev == c->ev[kEvBand]) g_band_chain_ev = nullptr;
    }
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->sc_host) (void)hipHostFree(c->sc_host);
    if (c->sc_out) (void)hipHostFree(c->sc_out);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    delete c;
}

int strk_count_loci_device(strk_ctx* ctx, const strk_batch* batch, const strk_params* params, int32_t* out_cn,
                           int32_t* out_score, int32_t* out_n_iters, int32_t* out_start, void* stream,
                           strk_stats* stats) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    return count_device(ctx, batch, params, out_cn, out_score, out_n_iters, out_start, static_cast<hipStream_t>(stream), stats);
}

int strk_submit_loci_device(strk_ctx* ctx, const strk_batch* batch, const strk_params* params, int32_t* out_cn,
                            int32_t* out_score, int32_t* out_n_iters, int32_t* out_start, void* stream) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    return submit_device(ctx, batch, params, out_cn, out_score, out_n_iters, out_start, static_cast<hipStream_t>(stream));
}

int strk_finish(strk_ctx* ctx, strk_stats* stats) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    return finish_device(ctx, stats);
}

int strk_count_loci(strk_ctx* ctx, const strk_batch* batch, const strk_params* params, int32_t* out_cn,
                    int32_t* out_score, int32_t* out_n_iters, int32_t* out_start, strk_stats* stats) {
    return strk_count_loci_dseqs(ctx, batch, nullptr, params, out_cn, out_score, out_n_iters, out_start, stats);
}

int strk_count_loci_dseqs(strk_ctx* ctx, const strk_batch* batch, const void* d_seqs, const strk_params* params, int32_t* out_cn,
                          int32_t* out_score, int32_t* out_n_iters, int32_t* out_start, strk_stats* stats) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    if (stats) memset(stats, 0, sizeof *stats);
    strk_batch d;
    int rc;
    if (d_seqs) {   // bases that are on the device already must be on THIS context's device (no peer access is set up)
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, d_seqs) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != ctx->device) {
            (void)hipGetLastError();
            return fail(STRK_E_INVALID, "d_seqs is not device memory of device %d (the context's)", ctx->device);
        }
    }
    if (!d_seqs) {   // host bases: large batches go through the pinned four-slot pipeline (strk_host_pipe.inc)
        bool taken = false;
        rc = count_loci_pipelined(ctx, batch, params, out_cn, out_score, out_n_iters, out_start, stats, &taken);
        if (taken) return rc;
    }
    if ((rc = upload_batch(ctx, batch, &d, nullptr, static_cast<const uint8_t*>(d_seqs)))) return rc;
    if (batch->n_reads == 0 || batch->n_loci == 0) return 0;
    if (!batch->est_cn) return fail(STRK_E_INVALID, "est_cn is NULL");
    const size_t nb = (size_t)batch->n_reads * 4;
    if ((rc = ctx->out_cn.ensure(nb)) || (rc = ctx->out_score.ensure(nb)) || (rc = ctx->out_n.ensure(nb)) ||
        (rc = ctx->out_start.ensure(nb)))
        return rc;
    rc = count_device(ctx, &d, params, ctx->out_cn.as<int32_t>(), ctx->out_score.as<int32_t>(), ctx->out_n.as<int32_t>(),
                      ctx->out_start.as<int32_t>(), nullptr, stats);
    if (rc && rc != STRK_E_EMPTY) return rc;
    const int rc_keep = rc;
    if (out_cn) HIP_TRY(hipMemcpy(out_cn, ctx->out_cn.p, nb, hipMemcpyDeviceToHost));
    if (out_score) HIP_TRY(hipMemcpy(out_score, ctx->out_score.p, nb, hipMemcpyDeviceToHost));
    if (out_n_iters) HIP_TRY(hipMemcpy(out_n_iters, ctx->out_n.p, nb, hipMemcpyDeviceToHost));
    if (out_start) HIP_TRY(hipMemcpy(out_start, ctx->out_start.p, nb, hipMemcpyDeviceToHost));
    return rc_keep;
}

int strk_repeat_count(strk_ctx* ctx, int32_t start_count, const uint8_t* tr, int32_t tr_len, const uint8_t* fl,
                      int32_t fl_len, const uint8_t* fr, int32_t fr_len, const uint8_t* motif, int32_t motif_len,
                      int32_t max_iters, int32_t local_search_range, int32_t step_size, int32_t* out_cn,
                      int32_t* out_score, int32_t* out_n_explored) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    if (tr_len < 0 || fl_len < 0 || fr_len < 0 || motif_len < 1) return fail(STRK_E_INVALID, "bad sequence length");
    if ((tr_len && !tr) || (fl_len && !fl) || (fr_len && !fr) || !motif) return fail(STRK_E_INVALID, "sequence pointer is NULL");
    const int32_t window = std::min(15, std::max(kDefaultWindow, local_search_range + step_size + 1));
    {
        int32_t cn = 0, sc = 0, n = 0;
        const int rf = scalar_fast(ctx, start_count, tr, tr_len, fl, fl_len, fr, fr_len, motif, motif_len, max_iters, local_search_range,
                                   step_size, window, &cn, &sc, &n);
        if (rf < 0) return rf;
        if (rf == 0) {
            if (out_cn) *out_cn = cn;
            if (out_score) *out_score = sc;
            if (out_n_explored) *out_n_explored = n;
            return 0;
        }
    }
    std::vector<uint8_t> seq((size_t)fl_len + tr_len + fr_len);
    if (fl_len) memcpy(seq.data(), fl, (size_t)fl_len);
    if (tr_len) memcpy(seq.data() + fl_len, tr, (size_t)tr_len);
    if (fr_len) memcpy(seq.data() + fl_len + tr_len, fr, (size_t)fr_len);
    const int64_t seq_off[2] = {0, (int64_t)seq.size()};
    const int32_t read_off[2] = {0, 1}, motif_off[2] = {0, motif_len};
    strk_batch b;
    b.n_reads = 1; b.n_loci = 1;
    b.seqs = seq.data(); b.seq_off = seq_off; b.nfl = &fl_len; b.ntr = &tr_len; b.nfr = &fr_len;
    b.est_cn = &start_count; b.read_off = read_off; b.motifs = motif; b.motif_off = motif_off;
    strk_params p;
    memset(&p, 0, sizeof p);
    p.max_iters = max_iters; p.local_search_range = local_search_range; p.step_size = step_size;
    p.tie_rule = STRK_TIE_FIRST; p.end_flags = STRK_SG_ALL; p.feedback = 0;
    p.window = window;
    int32_t cn = 0, sc = 0, n = 0, st = 0;
    const int rc = strk_count_loci(ctx, &b, &p, &cn, &sc, &n, &st, nullptr);
    if (rc) return rc;
    if (out_cn) *out_cn = cn;
    if (out_score) *out_score = sc;
    if (out_n_explored) *out_n_explored = n;
    return 0;
}

int strk_score_table(strk_ctx* ctx, const strk_batch* batch, const int32_t* lo, const int32_t* n,
                     const int64_t* table_off, int32_t end_flags, int32_t force_generic, int32_t* scores,
                     strk_stats* stats) {
    return score_table_impl(ctx, batch, lo, n, table_off, end_flags, force_generic, 0, scores, nullptr, stats);
}

int strk_score_ref_table(strk_ctx* ctx, const strk_batch* batch, const int32_t* lo, const int32_t* n,
                         const int64_t* table_off, int32_t force_generic, int32_t* scores, int32_t* end_query,
                         strk_stats* stats) {
    if (!end_query) return fail(STRK_E_INVALID, "end_query is NULL");
    return score_table_impl(ctx, batch, lo, n, table_off, STRK_DB_END_FREE, force_generic, 1, scores, end_query, stats);
}

int strk_ref_repeat_count(strk_ctx* ctx, int32_t start_count, const uint8_t* tr, int32_t tr_len, const uint8_t* fl,
                          int32_t fl_len, const uint8_t* fr, int32_t fr_len, const uint8_t* motif, int32_t motif_len,
                          int32_t ref_size, int32_t vcf_anchor_size, int32_t max_iters, int32_t local_search_range,
                          int32_t step_size, int32_t respect_coords, int32_t* out9) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    if (!out9) return fail(STRK_E_INVALID, "out9 is NULL");
    if (tr_len < 0 || fl_len < 0 || fr_len < 0 || motif_len < 1) return fail(STRK_E_INVALID, "bad sequence length");
    if ((tr_len && !tr) || (fl_len && !fl) || (fr_len && !fr) || !motif) return fail(STRK_E_INVALID, "sequence pointer is NULL");
    if (local_search_range < 0 || step_size < 1) return fail(STRK_E_INVALID, "local_search_range must be >= 0 and step_size >= 1");
    return ref_repeat_count_impl(ctx, start_count, tr, tr_len, fl, fl_len, fr, fr_len, motif, motif_len, ref_size,
                                 vcf_anchor_size, max_iters, local_search_range, step_size, respect_coords, out9);
}

int strk_realign(strk_ctx* ctx, int32_t n_pairs, const uint8_t* s1, const int64_t* s1_off, const uint8_t* s2,
                 const int64_t* s2_off, int32_t open, int32_t extend, int32_t gap_pref, int32_t* out_score,
                 int32_t* out_end_ref, int32_t* out_n_cigar, uint32_t* cigar, const int64_t* cigar_off, strk_stats* stats) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    if (ctx->pending) return fail(STRK_E_INVALID, "a submitted call is pending on this context");
    return realign_impl(ctx, n_pairs, s1, s1_off, s2, s2_off, open, extend, gap_pref, out_score, out_end_ref, out_n_cigar,
                        cigar, cigar_off, stats);
}

int strk_ref_repeat_count_batch(strk_ctx* ctx, int32_t n_loci, const int32_t* start_count, const uint8_t* seqs,
                                const int64_t* seq_off, const int32_t* nfl, const int32_t* ntr, const int32_t* nfr,
                                const uint8_t* motifs, const int32_t* motif_off, const int32_t* ref_size,
                                int32_t vcf_anchor_size, const int32_t* max_iters, const int32_t* local_search_range,
                                const int32_t* step_size, int32_t respect_coords, int32_t* out9) {
    if (!ctx) return fail(STRK_E_INVALID, "ctx is NULL");
    if (n_loci < 0) return fail(STRK_E_INVALID, "n_loci < 0");
    if (n_loci == 0) return 0;
    if (!start_count || !seqs || !seq_off || !nfl || !ntr || !nfr || !motifs || !motif_off || !ref_size || !max_iters ||
        !local_search_range || !step_size || !out9)
        return fail(STRK_E_INVALID, "NULL argument");
    std::vector<RefJob> jobs((size_t)n_loci);
    for (int32_t i = 0; i < n_loci; ++i) {
        const int32_t m = motif_off[i + 1] - motif_off[i];
        if (nfl[i] < 0 || ntr[i] < 0 || nfr[i] < 0 || m < 1 || seq_off[i + 1] - seq_off[i] != (int64_t)nfl[i] + ntr[i] + nfr[i])
            return fail(STRK_E_INVALID, "locus %d: bad sequence lengths", i);
        if (local_search_range[i] < 0 || step_size[i] < 1) return fail(STRK_E_INVALID, "locus %d: bad search schedule", i);
        const uint8_t* s0 = seqs + seq_off[i];
        ref_job_init(jobs[(size_t)i], start_count[i], s0 + nfl[i], ntr[i], s0, nfl[i], s0 + nfl[i] + ntr[i], nfr[i],
                     motifs + motif_off[i], m, ref_size[i], max_iters[i], local_search_range[i], step_size[i]);
    }
    return ref_repeat_count_batch_impl(ctx, jobs, vcf_anchor_size, respect_coords, out9);
}

int strk_realign_i16_flags(int32_t n_pairs, const int64_t* s1_off, const int64_t* s2_off, const int32_t* scores, int32_t* out_flags) {
    if (n_pairs < 0) return fail(STRK_E_INVALID, "n_pairs < 0");
    if (n_pairs == 0) return 0;
    if (!s1_off || !s2_off || !scores || !out_flags) return fail(STRK_E_INVALID, "NULL argument");
    constexpr int64_t kLimit = 32767 - 2;   // INT16_MAX less the largest matrix entry (align_matrix.py:15): parasail's head-room
    for (int32_t p = 0; p < n_pairs; ++p) {
        const int64_t n1 = s1_off[p + 1] - s1_off[p], n2 = s2_off[p + 1] - s2_off[p];
        if (n1 < 0 || n2 < 0) return fail(STRK_E_INVALID, "pair %d: negative length", p);
        int32_t f = 0;
        if (2 * std::min(n1, n2) > kLimit) f |= STRK_I16_CELL_MAY_SATURATE;   // no cell of an alignment exceeds 2 * min(rows, columns)
        if (scores[p] > kLimit) f |= STRK_I16_SCORE_SATURATES | STRK_I16_CELL_MAY_SATURATE;
        out_flags[p] = f;
    }
    return 0;
}

// ---- host-side front end (no device work, no context) -------------------------------------------------------------
static int64_t bam_scan_impl(const uint8_t* buf, int64_t n_bytes, int64_t first_rec, int64_t cap, int64_t* rec_off, int32_t* tid,
                             int32_t* pos, int32_t* end, int32_t* flag, int32_t* l_seq, int32_t* clip_l, int32_t* clip_r, int64_t* end_off) {
    if (!buf || n_bytes < 0 || first_rec < 0 || cap < 0) return fail(STRK_E_INVALID, "bad argument");
    if (cap > 0 && (!rec_off || !tid || !pos || !end || !flag || !l_seq || !clip_l || !clip_r)) return fail(STRK_E_INVALID, "NULL output array");
    int64_t off = first_rec, n = 0;
    while (off + 4 <= n_bytes) {
        strk_fe::Rec r;
        int64_t next = 0;
        if (end_off) {   // a piece of the stream: the last record may be cut off
            const int32_t block = strk_fe::rd_i32(buf + off);
            if (block >= 32 && off + 4 + block > n_bytes) break;
        }
        if (!strk_fe::parse_rec(buf, n_bytes, off, &r, &next)) return fail(STRK_E_INVALID, "malformed BAM record at byte %lld", (long long)off);
        if (n < cap) {
            int64_t ref_len = 0;
            int32_t cl = 0, cr = 0;
            for (int32_t i = 0; i < r.n_cigar; ++i) {
                const uint32_t c = strk_fe::rd_u32(r.cigar + 4 * (size_t)i), op = c & 15u;
                if (strk_fe::consumes_ref(op)) ref_len += c >> 4;
                if (op == 4 && i == 0) cl = (int32_t)(c >> 4);
                if (op == 4 && i == r.n_cigar - 1) cr = (int32_t)(c >> 4);
            }
            rec_off[n] = off; tid[n] = r.tid; pos[n] = r.pos; end[n] = (int32_t)(r.pos + ref_len); flag[n] = r.flag;
            l_seq[n] = r.l_seq; clip_l[n] = cl; clip_r[n] = cr;
        }
        ++n;
        off = next;
    }
    if (end_off) *end_off = off;
    return n;
}

int64_t strk_bam_scan(const uint8_t* buf, int64_t n_bytes, int64_t first_rec, int64_t cap, int64_t* rec_off, int32_t* tid,
                      int32_t* pos, int32_t* end, int32_t* flag, int32_t* l_seq, int32_t* clip_l, int32_t* clip_r) {
    return bam_scan_impl(buf, n_bytes, first_rec, cap, rec_off, tid, pos, end, flag, l_seq, clip_l, clip_r, nullptr);
}

int64_t strk_bam_scan_piece(const uint8_t* buf, int64_t n_bytes, int64_t first_rec, int64_t cap, int64_t* rec_off, int32_t* tid,
                            int32_t* pos, int32_t* end, int32_t* flag, int32_t* l_seq, int32_t* clip_l, int32_t* clip_r,
                            int64_t* end_off) {
    if (!end_off) return fail(STRK_E_INVALID, "end_off is NULL");
    return bam_scan_impl(buf, n_bytes, first_rec, cap, rec_off, tid, pos, end, flag, l_seq, clip_l, clip_r, end_off);
}

int64_t strk_bam_names(const uint8_t* buf, int64_t n_bytes, int64_t n, const int64_t* rec_off, uint8_t* out, int64_t out_cap,
                       int64_t* out_off) {
    if (!buf || n < 0 || (n > 0 && (!rec_off || !out_off))) return fail(STRK_E_INVALID, "bad argument");
    int64_t w = 0;
    if (out_off) out_off[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        strk_fe::Rec r;
        int64_t next = 0;
        if (!strk_fe::parse_rec(buf, n_bytes, rec_off[i], &r, &next)) return fail(STRK_E_INVALID, "malformed BAM record at byte %lld", (long long)rec_off[i]);
        const int64_t len = r.l_name > 0 ? r.l_name - 1 : 0;
        if (out) {
            if (w + len > out_cap) return fail(STRK_E_NOMEM, "name buffer too small");
            memcpy(out + w, r.name, (size_t)len);
        }
        w += len;
        out_off[i + 1] = w;
    }
    return w;
}

int strk_extract_reads(const uint8_t* buf, int64_t n_bytes, int32_t n_items, const int64_t* rec_off, const int64_t* coords,
                       const uint32_t* alt_cigar, const int64_t* alt_cigar_off, const int64_t* alt_start, int32_t flank_size,
                       int32_t min_avg_phred, int32_t wildcard_threshold, int32_t* status, int32_t* nfl, int32_t* ntr,
                       int32_t* nfr, uint8_t* seqs, int64_t seq_cap, int64_t* seq_off) {
    if (n_items < 0 || flank_size < 0) return fail(STRK_E_INVALID, "bad argument");
    if (n_items == 0) { if (seq_off) seq_off[0] = 0; return 0; }
    if (!buf || !rec_off || !coords || !status || !nfl || !ntr || !nfr || !seq_off) return fail(STRK_E_INVALID, "NULL argument");
    static const char kBases[] = "=ACMGRSVTWYHKDBN";
    // one thread per thousand items: starting threads costs more than a few hundred items do, and a caller that loads the next
    // block's records meanwhile (IndexedBam) needs the other cores
    const int nt = std::max(1, std::min<int>({host_cpus(), 32, n_items / 1024}));
    auto parallel = [&](auto&& body) {   // body(first item, last item): contiguous slices, one per thread
        std::vector<std::thread> th;
        const int32_t per = (n_items + nt - 1) / nt;
        for (int t = 1; t < nt; ++t)
            if (t * per < n_items) th.emplace_back(body, t * per, std::min(n_items, (t + 1) * per));
        body(0, std::min(n_items, per));
        for (auto& x : th) x.join();
    };
    // pass 1: where each read's flank | tract | flank lies (read positions a <= b <= c <= d), status, lengths
    std::vector<int64_t> cut((size_t)n_items * 2);   // a and b; c = b + ntr, d = c + nfr
    std::atomic<int> bad{-1};
    parallel([&](int32_t i0, int32_t i1) {
        strk_fe::Runs runs;
        for (int32_t it = i0; it < i1; ++it) {
            status[it] = 1; nfl[it] = ntr[it] = nfr[it] = 0;
            strk_fe::Rec r;
            int64_t next = 0;
            if (!strk_fe::parse_rec(buf, n_bytes, rec_off[it], &r, &next)) { bad.store(it); return; }
            const bool alt = alt_cigar && alt_cigar_off && alt_cigar_off[it + 1] > alt_cigar_off[it];
            if (alt) runs.build(reinterpret_cast<const uint8_t*>(alt_cigar + alt_cigar_off[it]), (int32_t)(alt_cigar_off[it + 1] - alt_cigar_off[it]), alt_start ? alt_start[it] : 0);
            else runs.build(r.cigar, r.n_cigar, r.pos);
            int64_t q[4];
            if (!strk_fe::read_coords(runs, coords[4 * (size_t)it], coords[4 * (size_t)it + 1], coords[4 * (size_t)it + 2], coords[4 * (size_t)it + 3], q)) continue;
            const int64_t b = q[1], c = q[2];
            const int64_t a = std::max(q[0], b - flank_size), d = std::min(q[3], c + flank_size);
            if (a < 0 || d > r.l_seq || a > b || b > c || c > d) continue;   // coordinates outside the read: incomplete
            const bool has_qual = !(r.l_seq > 0 && r.qual[0] == 0xFF);
            if (has_qual && c > b) {   // LowMeanBaseQual on the tract bases (call_locus.py:1099-1115)
                int64_t sum = 0;
                for (int64_t i = b; i < c; ++i) sum += r.qual[i];
                if ((double)sum / (double)(c - b) < (double)min_avg_phred) { status[it] = 2; continue; }
            }
            status[it] = 0;
            nfl[it] = (int32_t)(b - a); ntr[it] = (int32_t)(c - b); nfr[it] = (int32_t)(d - c);
            cut[2 * (size_t)it] = a;
        }
    });
    if (bad.load() >= 0) return fail(STRK_E_INVALID, "item %d: malformed BAM record", bad.load());
    int64_t w = 0;
    seq_off[0] = 0;
    for (int32_t it = 0; it < n_items; ++it) {
        w += (int64_t)nfl[it] + ntr[it] + nfr[it];
        seq_off[it + 1] = w;
    }
    if (!seqs) return 0;   // size query: seq_off[n_items] bytes are needed
    if (w > seq_cap) return fail(STRK_E_NOMEM, "sequence buffer too small (%lld < %lld)", (long long)seq_cap, (long long)w);
    // pass 2: bases (4 bit -> ASCII), low-quality bases -> 'X' (call_locus.py:79,1101-1106)
    parallel([&](int32_t i0, int32_t i1) {
        for (int32_t it = i0; it < i1; ++it) {
            if (status[it] != 0) continue;
            strk_fe::Rec r;
            int64_t next = 0;
            (void)strk_fe::parse_rec(buf, n_bytes, rec_off[it], &r, &next);
            const bool has_qual = !(r.l_seq > 0 && r.qual[0] == 0xFF);
            const int64_t a = cut[2 * (size_t)it], d = a + nfl[it] + ntr[it] + nfr[it];
            uint8_t* o = seqs + seq_off[it];
            for (int64_t i = a; i < d; ++i) {
                const uint8_t byte = r.seq[i >> 1];
                char ch = kBases[(i & 1) ? (byte & 15) : (byte >> 4)];
                if (has_qual && (int32_t)r.qual[i] <= wildcard_threshold) ch = 'X';
                *o++ = (uint8_t)ch;
            }
        }
    });
    return 0;
}

int64_t strk_bgzf_inflate_range(const uint8_t* comp, int64_t n_comp, int64_t coff, uint8_t* out, int64_t out_cap,
                                int64_t* next_coff, int32_t n_threads) {
    if (!comp || !out || !next_coff || n_comp < 0 || coff < 0 || coff > n_comp || out_cap < 0) return fail(STRK_E_INVALID, "bad argument");
    // walk the block headers from `coff` while the decompressed blocks still fit
    std::vector<strk_fe::BgzfBlock> blocks;
    int64_t off = coff, total = 0;
    while (off < n_comp) {
        std::vector<strk_fe::BgzfBlock> one;
        int64_t sz = 0;
        // the header of one block: reuse the indexer on a window that holds exactly this block
        if (off + 18 > n_comp) return fail(STRK_E_INVALID, "truncated BGZF block header at byte %lld", (long long)off);
        const uint8_t* p = comp + off;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return fail(STRK_E_INVALID, "not a BGZF block at byte %lld", (long long)off);
        const int xlen = strk_fe::rd_u16(p + 10);
        int bsize = -1;
        for (int64_t x = 12; x + 4 <= 12 + xlen && off + x + 4 <= n_comp;) {
            const int slen = strk_fe::rd_u16(p + x + 2);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2) bsize = strk_fe::rd_u16(p + x + 4);
            x += 4 + slen;
        }
        if (bsize < 0 || off + bsize + 1 > n_comp) return fail(STRK_E_INVALID, "truncated BGZF block at byte %lld", (long long)off);
        if (strk_fe::bgzf_index(p, bsize + 1, &one, &sz) || one.size() != 1) return fail(STRK_E_INVALID, "bad BGZF block at byte %lld", (long long)off);
        if (total + one[0].out_len > out_cap) break;
        one[0].in_off += off;
        one[0].out_off = total;
        total += one[0].out_len;
        blocks.push_back(one[0]);
        off += bsize + 1;
    }
    *next_coff = off;
    const int nt = std::max(1, std::min<int>(n_threads > 0 ? n_threads : host_cpus(), 32));
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(16);
            if (i >= blocks.size()) return;
            for (size_t k = i; k < std::min(blocks.size(), i + 16); ++k)
                if (!strk_fe::bgzf_inflate_block(comp, blocks[k], out)) bad.store(1);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt && (size_t)t * 16 < blocks.size() + 16; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (bad.load()) return fail(STRK_E_INVALID, "corrupt BGZF block (inflate or CRC failed)");
    return total;
}

int64_t strk_bgzf_inflate(const uint8_t* comp, int64_t n_comp, uint8_t* out, int64_t out_cap, int32_t n_threads) {
    if (!comp || n_comp < 0) return fail(STRK_E_INVALID, "bad argument");
    std::vector<strk_fe::BgzfBlock> blocks;
    int64_t total = 0;
    if (strk_fe::bgzf_index(comp, n_comp, &blocks, &total)) return fail(STRK_E_INVALID, "not a BGZF stream (or truncated)");
    if (!out) return total;   // size query
    if (out_cap < total) return fail(STRK_E_NOMEM, "output buffer too small (%lld < %lld)", (long long)out_cap, (long long)total);
    const int nt = std::max(1, std::min<int>(n_threads > 0 ? n_threads : host_cpus(), 32));
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(16);
            if (i >= blocks.size()) return;
            for (size_t k = i; k < std::min(blocks.size(), i + 16); ++k)
                if (!strk_fe::bgzf_inflate_block(comp, blocks[k], out)) bad.store(1);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (bad.load()) return fail(STRK_E_INVALID, "corrupt BGZF block (inflate or CRC failed)");
    return total;
}

}  // extern "C"

#include "strk_dbam.inc"

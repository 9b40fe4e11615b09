// api_common.hpp -- helpers shared by the C-ABI translation units.
#pragma once

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "encode_kernels.hpp"
#include "regex_device.hpp"
#include "runtime.hpp"
#include "tables.hpp"

// Handle of RegexSplit (shared by api_encode.cpp and the fused WordPiece path in api_ops.cpp).
struct ovtk_regex_split {
    int device = 0;
    ovtk::SplitDev dev{};  // dev.kind: a hand-written scanner, or kSplitGeneral: the compiled DFA below
    int mode = 1;  // 0 removed, 1 isolated, 2 merged-with-previous, 3 merged-with-next
    bool invert = false;
    int max_splits = -1;
    ovtk::RegexDev regex{};
    ovtk::DevBuf r_trans, r_ascii, r_index, r_blocks, r_ctx;
};

namespace ovtk {

inline int use_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(OVTK_E_HIP, "no HIP device is available; this library has no CPU execution path");
    if (device < 0 || device >= n) return set_error(OVTK_E_ARG, "device ordinal out of range");
    OVTK_HIP(hipSetDevice(device));
    return OVTK_OK;
}

inline StringsView view_of(const ovtk_strings& s) { return StringsView{s.begins, s.ends, s.chars, s.n}; }

// Entries the piece store (tables.hpp) of the handles created from now on may take unless their create parameters say otherwise
// (ovtk_bpe_params::memo_store); 0: no store.  The reference's attribute list has no room for it (cache_capacity keeps its meaning:
// 0 = no memo at all).  Process-wide and atomic (as a thread_local it was silently ignored by handles created on
// another thread than the one that set it).
inline std::atomic<int64_t>& memo_store_entries() {
    static std::atomic<int64_t> v{1048576};
    return v;
}
// what a create call's `memo_store` parameter means: 0 = the default above, < 0 = none
inline int64_t resolve_memo_store(int64_t param) { return param == 0 ? memo_store_entries().load(std::memory_order_relaxed) : (param < 0 ? 0 : param); }
// A handle's piece store: the table (zeroed), its room counter, the device view.  A piece's two candidate slots are the
// halves of one 128-byte line and nothing is relocated on the device: the table is kept below a third full (an insert finds
// its line taken a few times in a hundred then).  No more than a few entries per vocabulary token: a 3 000-token test vocabulary
// does not need 32 MiB of table.
// entries a handle's store takes for a requested capacity (never more than four per vocabulary token)
inline int64_t piece_store_capacity(int64_t vocab_n, int64_t entries) {
    return std::max<int64_t>(0, std::min<int64_t>({entries, int64_t(1) << 22, std::max<int64_t>(8192, 4 * vocab_n)}));
}
inline int alloc_piece_store(DevBuf& table, DevBuf& room, int64_t vocab_n, bool narrow, PieceStoreDev& dev, int32_t& capacity, int64_t entries) {
    dev = PieceStoreDev{nullptr, 30, nullptr, 0};
    capacity = 0;
    const int64_t want = piece_store_capacity(vocab_n, entries);
    if (want <= 0) return OVTK_OK;
    const uint32_t slots = std::max<uint32_t>(1024, pow2_at_least(uint64_t(want) * 3));
    if (int rc = table.ensure(size_t(slots) * sizeof(StoreEntry))) return rc;
    OVTK_HIP(hipMemset(table.as<void>(), 0, size_t(slots) * sizeof(StoreEntry)));
    const int32_t sroom = int32_t(want);
    if (int rc = room.upload(&sroom, sizeof sroom)) return rc;
    OVTK_HIP(hipStreamSynchronize(nullptr));
    capacity = sroom;
    dev = PieceStoreDev{table.as<StoreEntry>(), 32u - uint32_t(log2u(slots / 2)), room.as<int32_t>(), narrow ? 1 : 0};  // (shift of the LINE index)
    return OVTK_OK;
}

inline int check_rows(const ovtk_ragged_strings* in) {
    if (!in) return set_error(OVTK_E_ARG, "null input");
    if (in->n_rows < 0 || in->strings.n < 0 || in->strings.n_chars < 0) return set_error(OVTK_E_ARG, "negative size");
    if (in->n_rows >= INT32_MAX || in->strings.n >= INT32_MAX || in->strings.n_chars >= INT32_MAX)
        return set_error(OVTK_E_ARG, "tensor sizes must fit int32 offsets (the reference's begins/ends are i32)");
    return OVTK_OK;
}

// Brings the input ragged string tensor to the device (or wraps device pointers).
inline int stage_input(Workspace& ws, const ovtk_ragged_strings* in, const uint8_t* skips, int mem, hipStream_t s, RowsIn& d) {
    d.n_rows = int32_t(in->n_rows);
    d.n_strings = int32_t(in->strings.n);
    d.n_chars = in->strings.n_chars;
    if (mem == OVTK_MEM_DEVICE) {
        d.ragged_begins = in->ragged_begins;
        d.ragged_ends = in->ragged_ends;
        d.begins = in->strings.begins;
        d.ends = in->strings.ends;
        d.chars = in->strings.chars;
        d.skips = skips;
        return OVTK_OK;
    }
    if (mem != OVTK_MEM_HOST) return set_error(OVTK_E_ARG, "mem must be OVTK_MEM_HOST or OVTK_MEM_DEVICE");
    int e = 0;
    e = e ? e : ws.in_rb.upload(in->ragged_begins, size_t(in->n_rows) * 4, s);
    e = e ? e : ws.in_re.upload(in->ragged_ends, size_t(in->n_rows) * 4, s);
    e = e ? e : ws.in_begins.upload(in->strings.begins, size_t(in->strings.n) * 4, s);
    e = e ? e : ws.in_ends.upload(in->strings.ends, size_t(in->strings.n) * 4, s);
    e = e ? e : ws.in_chars.upload(in->strings.chars, size_t(in->strings.n_chars), s);
    if (skips) e = e ? e : ws.in_skips.upload(skips, size_t(in->strings.n), s);
    if (e) return e;
    d.ragged_begins = ws.in_rb.as<int32_t>();
    d.ragged_ends = ws.in_re.as<int32_t>();
    d.begins = ws.in_begins.as<int32_t>();
    d.ends = ws.in_ends.as<int32_t>();
    d.chars = ws.in_chars.as<uint8_t>();
    d.skips = skips ? ws.in_skips.as<uint8_t>() : nullptr;
    return OVTK_OK;
}

inline int grid_for_rows(int n_rows) { return std::max(1, (n_rows + kWavesPerBlock - 1) / kWavesPerBlock); }

inline int finish_status(Workspace& ws, hipStream_t s) {
    OVTK_HIP(hipMemcpyAsync(ws.host_status, ws.status.as<RunStatus>(), sizeof(RunStatus), hipMemcpyDeviceToHost, s));
    OVTK_HIP(hipStreamSynchronize(s));
    ws.marks.settled();
    Profiler::get().resolve(ws.marks);
    OVTK_HIP(hipGetLastError());
    return OVTK_OK;
}


inline int grid_for_elems(long long n) { return int(std::max<long long>(1, std::min<long long>((n + kBlockThreads - 1) / kBlockThreads, 1 << 20))); }

// Device-side destination of a host/device output buffer of `bytes` bytes: the caller's pointer
// (OVTK_MEM_DEVICE) or a workspace buffer that finish_output() copies back (OVTK_MEM_HOST).
template <typename T>
inline int out_target(DevBuf& stage, T* user, size_t bytes, int mem, T** dev) {
    if (mem == OVTK_MEM_DEVICE) { *dev = user; return OVTK_OK; }
    if (int rc = stage.ensure(bytes)) return rc;
    *dev = stage.as<T>();
    return OVTK_OK;
}
// The device-side address of a caller's PINNED host buffer (hipHostMalloc / hipHostRegister: mapped into the device's
// address space), or nullptr for pageable memory.  The last kernel of an encode writes its outputs straight through it.
// On this runtime (ROCm 7) a D2H hipMemcpyAsync is a copy KERNEL (rocprofv3 --memory-copy-trace shows H2D only), which the
// host can issue only after it has read the id count -- one finish() at a time, each blocked for the length of its copy:
// the pinned-buffer pipeline settles at H2D + D2H per batch (1.22 ms for config 2; only its first ~30 batches run at the
// 0.8 ms that two-way PCIe allows).  compact_kernel's stores over PCIe need neither the copy nor the host: 0.78-0.80 ms per
// batch in the steady state (tools/e2e_age_probe.py).  Tried on the way and dropped: copy streams of the library's own, one
// per direction (no better with the copy kernel, and two more streams move every stream of the process to other hardware
// queues: the device-resident pipeline went from 0.165 to 0.22 ms per step).
// `bytes`: the kernels will store that many bytes from `user` on -- the LAST byte must belong to the same pinned mapping (a
// buffer pinned only in part would otherwise be a GPU page fault instead of the staged copy).  The mapping of a buffer
// registered without hipHostRegisterPortable is valid on the device it was registered on: the check below queries the
// attributes with the run's device current (the callers have set it), and takes the device-side pointer that query returns.
inline void* mapped_host_pointer(const void* user, size_t bytes = 1) {
    if (!user) return nullptr;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, user) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory is reported as an error by some runtimes
        return nullptr;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    if (bytes > 1) {
        hipPointerAttribute_t z{};
        const char* last = static_cast<const char*>(user) + (bytes - 1);
        if (hipPointerGetAttributes(&z, last) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        // the same registration maps both ends: host and device addresses advance together
        if (z.type != hipMemoryTypeHost || !z.devicePointer ||
            static_cast<const char*>(z.devicePointer) - static_cast<const char*>(a.devicePointer) != std::ptrdiff_t(bytes - 1))
            return nullptr;
    }
    return a.devicePointer;
}
inline int copy_back(void* user, const void* dev, size_t bytes, int mem, hipStream_t s) {
    if (mem == OVTK_MEM_DEVICE || bytes == 0) return OVTK_OK;
    OVTK_HIP(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, s));
    return OVTK_OK;
}
// Input counterpart: device pointer of an input of `bytes` bytes.
template <typename T>
inline int in_source(DevBuf& stage, const T* user, size_t bytes, int mem, hipStream_t s, const T** dev) {
    if (mem == OVTK_MEM_DEVICE) { *dev = user; return OVTK_OK; }
    if (int rc = stage.upload(user, bytes, s)) return rc;
    *dev = stage.as<T>();
    return OVTK_OK;
}

// Device-wide exclusive scan (scan_kernels.hpp): three launches on `s`.
template <class LenF, class ApplyF, class Fin>
inline void launch_scan(std::vector<Profiler::Mark>& marks, const char* tag, hipStream_t s, long long n, LenF len,
                        ApplyF apply, Fin fin, long long* tiles, const RunStatus* st, uint32_t skip_flags) {
    const int n_tiles = int((n + kTileElems - 1) / kTileElems);
    if (n_tiles > 0) OVTK_LAUNCH(marks, tag, tile_reduce_kernel<LenF>, n_tiles, kTileThreads, s, n, len, tiles);
    OVTK_LAUNCH(marks, tag, tile_scan_kernel<Fin>, 1, kScanThreads, s, n_tiles, tiles, fin);
    if (n_tiles > 0)
        OVTK_LAUNCH(marks, tag, (tile_apply_kernel<LenF, ApplyF>), n_tiles, kTileThreads, s, n, len,
                    (const long long*)tiles, apply, st, skip_flags);
}
inline size_t scan_tiles_bytes(long long n) { return size_t((n + kTileElems - 1) / kTileElems + 1) * sizeof(long long); }

constexpr int kTicketBlocks = 256;  // blocks of the launches that end with a "last block done" ticket
// Persistent grids: exactly the blocks that are resident at the kernel's occupancy (a larger grid runs its surplus as a
// second, poorly filled round: every block carries the same share of the rows); rows / list entries are strided.
// The occupancy API over-reports by one block per CU for SGPR-heavy kernels (MI355X_MICROARCH.md "Residency"), hence
// the cap at 6 (100+ SGPRs).
template <class Kernel>
inline int resident_blocks_per_cu(Kernel kernel, int cap = 6) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlockThreads, 0) != hipSuccess || n <= 0) n = 4;
    return std::min(n, cap);
}
inline int grid_rows(int device, int n_rows, int blocks_per_cu) {
    return std::max(1, std::min((n_rows + kWavesPerBlock - 1) / kWavesPerBlock, device_cu_count(device) * blocks_per_cu));
}
inline int grid_lookup(int device, int n_rows) { return grid_rows(device, n_rows, 6); }
// Grid of the kernels whose waves own CONSECUTIVE rows (lookup_span_kernel, lookup_rows_kernel): the persistent grid with
// ceil(n_rows / waves) rows per wave -- or, where that would be more than the 64 a wave's lanes can hold headers for (hundreds of
// thousands of short rows), 64 rows per wave and as many blocks as that takes: the surplus runs as later rounds.
inline int rows_grid(int n_rows, int persistent_grid, int32_t& rows_per_wave) {
    static const int rows_env = [] { const char* e = std::getenv("OVTK_SPAN_ROWS"); return e ? std::atoi(e) : 0; }();   // (measurements: rows per wave, any grid)
    if (rows_env >= 1 && rows_env <= kWave) {
        rows_per_wave = rows_env;
        return std::max(1, (n_rows + rows_env * kWavesPerBlock - 1) / (rows_env * kWavesPerBlock));
    }
    const int waves = persistent_grid * kWavesPerBlock;
    rows_per_wave = (n_rows + waves - 1) / waves;
    // A multiple of compact_kernel's work item (kCompactRows = 4 rows) where a wave owns more than that: an item whose rows were staged by
    // ONE wave is one stretch of the staging buffer (compact_flat); at 13 rows per wave three items in ten straddled two waves and went row
    // by row.  (Config 2: 16 rows per wave, 4 096 of the 5 120 resident waves have rows -- lookup_span_kernel alone 58.2 -> 58.5 us, every
    // block of a wave's chain full.)
    if (rows_per_wave > 4) rows_per_wave = (rows_per_wave + 3) & ~3;
    if (rows_per_wave <= kWave) {   // (no blocks whose waves have no rows)
        const int rows_per_block = rows_per_wave * kWavesPerBlock;
        return std::max(1, std::min(persistent_grid, (n_rows + rows_per_block - 1) / rows_per_block));
    }
    rows_per_wave = kWave;
    return (n_rows + kWave * kWavesPerBlock - 1) / (kWave * kWavesPerBlock);
}
// Blocks per shard of the kernels that work through the deferred list (merge_kernel, wordpiece_deferred_kernel): the
// resident number, but no more than the list can give work to -- a piece has at least one byte, a wave takes 64 pieces.
// Every block draws a "last block done" ticket from one counter (~90 atomics per microsecond on one address), so a
// 1 000-block grid costs a 32-row batch 11 us of tickets alone.
inline int grid_deferred_per_shard(long long n_chars, long long n_strings, int resident_per_shard) {
    const long long most_batches = (n_chars + n_strings + (long long)kShards * kWave - 1) / ((long long)kShards * kWave);
    const long long blocks = (2 * most_batches + kWavesPerBlock - 1) / kWavesPerBlock;  // x2: shards fill unevenly when few blocks feed them
    return int(std::max<long long>(1, std::min<long long>(blocks, resident_per_shard)));
}

// ... and, the short path (EncodeWork::span_sums: the list holds only what is in neither table), no more than a list of about
// `hint` pieces can use -- the kernels' fixed cost follows their grid (a launch that finds 13 words: 18 us on the full grid).  The
// batches are strided over whatever grid there is, so a list that turns out longer is worked through all the same, by fewer waves.
inline int grid_deferred_hinted(int full_per_shard, int span_sums, int hint) {
    if (!span_sums || hint < 0) return full_per_shard;
    const long long blocks = (4ll * hint + 1024 + (long long)kShards * kBlockThreads - 1) / ((long long)kShards * kBlockThreads);
    return int(std::max<long long>(std::min<long long>(4, full_per_shard), std::min<long long>(blocks, full_per_shard)));
}

// The "ragged strings in -> ragged i32 out" pipeline shared by BPETokenizer, the fused encode and
// WordpieceTokenizer: prep (validation + per-wave staging arenas) -> middle(ws, d_in, w, grid) (the op's kernels: ids
// into staging, per-row counts; the first one must be launched with `grid` blocks, the geometry prep summed over) ->
// count_scan (final offsets) -> compact.  Workspace overflows reported
// by the kernels are handled by growing the buffer and running again.
// self_alloc: the middle's first kernel (lookup_kernel) validates the offsets and takes its staging from bump allocators
// itself, so no prep launch is needed.
//
// A run is an object so that a call can be split in two (ovtk_encode_enqueue / ovtk_encode_finish): start() stages the
// inputs and launches one attempt without waiting; finish() waits for that attempt's event (not for whatever the
// caller put on the stream afterwards), repeats it while a workspace was too small, and reports.
// ovtk_set_row_tickets(): 0 = every wave owns a fixed share of the rows; n > 0 = rows are handed out n at a time.
inline std::atomic<int>& row_tickets() {
    static std::atomic<int> v{0};
    return v;
}

// ovtk_set_short_path(): 0 = never, 1 (default) = where a handle's last calls say it will work, 2 = every eligible call tries.
inline std::atomic<int>& short_path_mode() {
    static std::atomic<int> v{1};
    return v;
}

// Calls for which a kernel of the middle is still launched after a call that had work for it (the handles' predictors: api_encode.cpp
// ovtk_bpe).  A wrong guess costs one more round of launches (~0.1 ms), a needless launch a few microseconds: four calls without work
// and the kernel is left out -- the driver's bench line (8 priming + 5 warm-up calls in front of 20 timed ones) then times the steady form.
constexpr int kShortPathKeep = 4;
// ovtk_short_path_stats(): calls launched as span -> compact / of them, calls that needed no other kernel.
struct ShortPathCounts { std::atomic<int64_t> tried{0}, exact{0}; };
inline ShortPathCounts& short_path_counts() {
    static ShortPathCounts c;
    return c;
}

struct PendingRun {
    virtual ~PendingRun() = default;
    virtual int finish(ovtk_ragged_i32_out* out) = 0;
};
struct PendingStrings {  // a call in flight whose result is a strings tensor (ovtk_detokenize_enqueue)
    virtual ~PendingStrings() = default;
    virtual int finish(ovtk_strings_out* out) = 0;
};
}  // namespace ovtk
// A call in flight (ovtk_encode_enqueue / ovtk_wordpiece_encode_enqueue -> ovtk_encode_finish;
// ovtk_detokenize_enqueue -> ovtk_detokenize_finish).
struct ovtk_pending {
    std::unique_ptr<ovtk::PendingRun> run;  // empty: nothing was launched, `out` was complete at enqueue
    ovtk_ragged_i32_out out{};
    std::unique_ptr<ovtk::PendingStrings> strings;
    ovtk_strings_out strings_out{};
    std::shared_ptr<int32_t> dense_width;   // ovtk_encode_dense_enqueue: the row width, known when the run has finished
};
namespace ovtk {

// Up to this many rows the last block of merge_kernel sums the row counts itself (one block reads 4 bytes per row);
// larger batches keep the parallel count_scan_kernel.
constexpr int kFoldTailRows = 1 << 18;
constexpr int kTileSumTiles = 1024;   // (65 536 rows)
// A batch this small is one launch of one block (encode_small_kernel): BASELINE config 1 is 32 rows / 4 KB.
constexpr int kSmallRows = 256;
constexpr int64_t kSmallChars = 64 << 10;


template <class Middle>
class RowsRun final : public PendingRun {
public:
    RowsRun(int device, const char* op, const ovtk_ragged_strings* in, const uint8_t* skips, int mul,
            const ovtk_ragged_i32_out* out, int mem, hipStream_t s, Middle middle, bool self_alloc, int blocks_per_cu,
            bool tail_in_middle)
        : device_(device), op_(op), in_(*in), skips_(skips), mul_(mul), out_(*out), mem_(mem), in_mem_(mem), s_(s),
          middle_(std::move(middle)), self_alloc_(self_alloc), blocks_per_cu_(blocks_per_cu), ws_(device),
          fold_tail_(tail_in_middle) {}

    // The input tensors live in device memory whatever `mem` says about the outputs (pieces produced on the device by
    // a preceding split; `keep` = what owns them, released with this run).
    void input_on_device(std::shared_ptr<void> keep) {
        in_mem_ = OVTK_MEM_DEVICE;
        keep_ = std::move(keep);
    }
    // The middle knows the one-launch form of a small batch (EncodeWork::small, encode_small_kernel).
    void enable_small() { small_ok_ = true; }
    // Called with the status block of the attempt that completed the run (finish(), the caller's thread).
    void on_status(std::function<void(const RunStatus&)> f) { on_status_ = std::move(f); }
    // Kernels of another workspace run in front of this run's on the same stream (a split into device buffers): once this
    // run's event has completed, so have they.
    void also_settles(std::shared_ptr<WorkspaceLease> other) { others_.push_back(std::move(other)); }
    // Asked once this run's event has completed, before its own status is looked at: what a stage in front of it (another
    // workspace's kernels on the same stream) has to say; non-zero ends finish() with that code.
    void front_check(std::function<int()> f) { front_check_ = std::move(f); }
    // The middle's first kernel is lookup_span_kernel and does the middle's bookkeeping itself (EncodeWork::span_sums, span_kernel.hpp
    // "the short path"); of the kernels behind it only those are launched that the handle expects to find work -- lookup_kernel<kFused>
    // for left-over rows, merge_kernel / wordpiece_deferred_kernel for pieces that are in neither table.  What was left out and was
    // needed after all follows from finish() (compact_kernel wrote nothing then).
    // merge_hint: pieces the handle's last call left for merge_kernel / wordpiece_deferred_kernel (< 0: unknown) -- sizes that launch.
    void enable_short_path(bool expect_pending, bool expect_merge, int64_t merge_hint) {
        short_ok_ = true;
        expect_pending_ = expect_pending;
        expect_merge_ = expect_merge;
        merge_hint_ = merge_hint;
    }
    void enable_stage16() { stage16_ = true; }   // the middle's kernels write / read the staging entries as u16 (EncodeWork::stage16)
    // The middle's first kernel takes staging for ALL its rows before it knows which of them it will leave to the kernel behind it
    // (lookup_span_kernel: one reservation per wave), and that kernel takes its own: room for both, or every call with left-over
    // rows would overflow, grow and run twice.
    void stage_twice() { stage_twice_ = true; }
    // The result leaves in the row-shard exchange's wire form (device memory) instead of begins / ends / ids.
    void output_to_wire(const WireSink& wire) {
        wire_ = wire;
        small_ok_ = false;
    }
    // The result leaves as input_ids / attention_mask [n_rows, T] (device memory): DenseSink, the graph's tail in compact_kernel.
    void output_dense(const DenseSink& dense) {
        dense_ = dense;
        dense_on_ = true;
        small_ok_ = false;
    }

    int start() {
        if (!ws_->host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
        if (!ws_->done) OVTK_HIP(hipEventCreateWithFlags(&ws_->done, hipEventDisableTiming));
        if (int rc = stage_input(*ws_.ws, &in_, skips_, in_mem_, s_, d_in_)) return rc;
        n_rows_ = d_in_.n_rows;
        stage_cap_ = std::min<int64_t>((in_.strings.n_chars + in_.strings.n) * mul_, INT32_MAX - 1);
        shard_cap_ = std::max<int64_t>({4096, (in_.strings.n_chars / 16 + in_.strings.n / 4) / kShards,
                                        int64_t(ws_->deferred.size() / sizeof(DeferredPiece) / kShards)});
        exact_cap_ = std::max<int64_t>(4096, int64_t(ws_->exact.size() / sizeof(ExactPiece)));
        scratch_cap_ = std::max<int64_t>(ws_->scratch.size(), int64_t(16) << 20);
        if (!wire_.hdr && !dense_on_) {
            if (mem_ == OVTK_MEM_HOST) {  // pinned output buffers are written by the kernels themselves
                void* pb = mapped_host_pointer(out_.begins, size_t(n_rows_) * 4);
                void* pe = mapped_host_pointer(out_.ends, size_t(n_rows_) * 4);
                void* pd = mapped_host_pointer(out_.data, size_t(std::max<int64_t>(out_.data_capacity, 1)) * 4);
                if (pb && pe && pd) {
                    d_begins_ = static_cast<int32_t*>(pb);
                    d_ends_ = static_cast<int32_t*>(pe);
                    d_ids_ = static_cast<int32_t*>(pd);
                    direct_out_ = true;
                }
            }
            if (!direct_out_) {
                if (int rc = out_target(ws_->out_a, out_.begins, size_t(n_rows_) * 4, mem_, &d_begins_)) return rc;
                if (int rc = out_target(ws_->out_b, out_.ends, size_t(n_rows_) * 4, mem_, &d_ends_)) return rc;
                if (int rc = out_target(ws_->out_c, out_.data, size_t(out_.data_capacity) * 4, mem_, &d_ids_)) return rc;
            }
        }
        grid_ = grid_rows(device_, n_rows_, blocks_per_cu_);
        n_tiles_ = (n_rows_ + kRowTile - 1) / kRowTile;
        if (stage_twice_) stage_cap_ = std::min<int64_t>(stage_cap_ * 2, INT32_MAX - 1);
        if (self_alloc_) {  // every wave may leave one chunk partly unused -- the waves that are LAUNCHED: lookup_rows_kernel's grid grows
                           // beyond the persistent one when a wave would own more than 64 rows (rows_grid(); ADVICE r04: with the persistent
                           // grid's slack a batch of a million short rows overflowed at its first attempt and ran twice)
            const int64_t waves = std::max<int64_t>(int64_t(grid_) * kWavesPerBlock, (int64_t(n_rows_) + kWave - 1) / kWave);
            stage_cap_ = std::min<int64_t>(stage_cap_ + (waves + kShards) * kStageChunk, INT32_MAX - 1);
        }
        small_ = small_ok_ && self_alloc_ && fold_tail_ && n_rows_ <= kSmallRows && in_.strings.n_chars <= kSmallChars;
        return launch();
    }

    int finish(ovtk_ragged_i32_out* out) override {
        OVTK_HIP(hipSetDevice(device_));  // the retry path allocates and launches: on this run's device, whatever is current
        for (int attempt = 0; attempt < 6; ++attempt) {
            OVTK_HIP(hipEventSynchronize(ws_->done));
            ws_->marks.settled();
            Profiler::get().resolve(ws_->marks);
            for (auto& o : others_) {
                o->ws->marks.settled();
                Profiler::get().resolve(o->ws->marks);
            }
            OVTK_HIP(hipGetLastError());
            if (front_check_)
                if (int rc = front_check_()) return rc;
            RunStatus& st = *ws_->host_status;
            if (phase_ != 0) {   // (the short path: every entry of the deferred list is a piece in neither table)
                long long unresolved = 0;
                for (int k = 0; k < kShards; ++k) unresolved += st.shard_count[k * kCounterStride];
                st.n_unresolved = int32_t(std::min<long long>(unresolved, INT32_MAX));
            }
            static const bool debug_status = std::getenv("OVTK_DEBUG_STATUS") != nullptr;
            if (debug_status) {   // (a debugging aid: what the kernels reported, one line per attempt)
                long long deferred = 0;
                for (int k = 0; k < kShards; ++k) deferred += st.shard_count[k * kCounterStride];
                std::fprintf(stderr, "[ovtk] %s rows %d phase %d flags %#x n_out %d deferred %lld pending %d unresolved %d exact-list %d store probes %d hits %d\n", op_.c_str(),
                             n_rows_, phase_, st.flags, st.n_out, deferred, st.n_pending, st.n_unresolved, st.n_exact, st.n_store_probe, st.n_store_hit);
            }
            if (st.flags & kFlagDidNotRun) return set_error(OVTK_E_HIP, op_ + ": the call's last kernel did not run (an earlier launch failed)");
            if (pending_clean_) {   // the one-launch kernel has run: it left the status block zeroed behind itself
                ws_->clean_status = pending_clean_;
                ws_->clean_bytes = pending_clean_bytes_;
                ws_->clean_after_lease = ws_->lease_count;
                pending_clean_ = nullptr;
            }
            if (pending_zeroed_) {   // compact_kernel has run: it zeroed the workspace's other status block
                ws_->zeroed_status = pending_zeroed_;
                ws_->zeroed_bytes = pending_zeroed_bytes_;
                ws_->zeroed_after_lease = ws_->lease_count;
                pending_zeroed_ = nullptr;
            }
            if (st.flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
            if (st.flags & kFlagStageOverflow) {
                const int64_t need = self_alloc_ ? stage_cap_ * 2 : int64_t(st.stage_need);  // the allocators do not know the total
                if (need >= INT32_MAX - 1)
                    return set_error(OVTK_E_UNSUPPORTED, "batch needs more than 2^31 staging entries; split the call");
                stage_cap_ = need;
            } else if (st.flags & kFlagDeferOverflow) {
                int64_t most = 0;
                for (int k = 0; k < kShards; ++k) most = std::max<int64_t>(most, st.shard_count[k * kCounterStride]);
                shard_cap_ = most + most / 8 + 256;
            } else if (st.flags & kFlagExactOverflow) {
                exact_cap_ = int64_t(st.n_exact) + 256;
            } else if (st.flags & kFlagScratchOverflow) {
                scratch_cap_ = std::max<int64_t>(scratch_cap_ * 2, int64_t(st.scratch_used) + (1 << 20));
                if (scratch_cap_ > (int64_t(3) << 30))
                    return set_error(OVTK_E_UNSUPPORTED, "exact-path scratch would exceed 3 GiB; split the call");
            } else if (st.flags & kFlagTailPending) {
                fold_tail_ = false;  // more exact pieces than the folded tail takes: once more with their own launches
                small_ = false;
            } else if (phase_ == 1 && (((w_.skip_mask & kSkipPending) && st.n_pending > 0) || ((w_.skip_mask & kSkipMerge) && st.n_unresolved > 0))) {
                short_path_counts().tried.fetch_add(1, std::memory_order_relaxed);
                // the short path: rows or pieces were left to a kernel that was not launched, compact_kernel wrote nothing -- that kernel
                // after all, and compact_kernel again
                if (int rc = launch_phase2(st)) return rc;
                continue;
            } else {
                if (phase_ == 1 && w_.skip_mask) {
                    short_path_counts().tried.fetch_add(1, std::memory_order_relaxed);
                    short_path_counts().exact.fetch_add(1, std::memory_order_relaxed);
                }
                out->n_data = st.n_out;
                if (on_status_) {
                    RunStatus seen = st;
                    seen.short_path = force_long_ ? 3 : phase_;   // (3: started over the long way -- rows were left over behind a merge_kernel that had run)
                    on_status_(seen);
                }
                if (st.flags & kFlagOutCapacity)
                    return set_error(OVTK_E_CAPACITY, dense_on_ ? op_ + ": dense outputs too small (row width " + std::to_string(st.width) + ")"
                                                                : op_ + ": output ids buffer too small (" + std::to_string(st.n_out) +
                                                                      " ids, capacity " + std::to_string(out_.data_capacity) + ")");
                if (wire_.hdr || dense_on_) return OVTK_OK;  // (device memory: the wire / the dense tensors are complete)
                if (direct_out_) return OVTK_OK;  // the kernels wrote the caller's pinned buffers
                if (int rc = copy_back(out_.begins, d_begins_, size_t(n_rows_) * 4, mem_, s_)) return rc;
                if (int rc = copy_back(out_.ends, d_ends_, size_t(n_rows_) * 4, mem_, s_)) return rc;
                if (int rc = copy_back(out_.data, d_ids_, size_t(st.n_out) * 4, mem_, s_)) return rc;
                if (mem_ == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s_));
                return OVTK_OK;
            }
            small_ = false;                    // (whatever it was: the repeat takes the ordinary launches)
            if (int rc = launch()) return rc;  // a workspace was too small: once more with the size the kernels asked for
        }
        return set_error(OVTK_E_HIP, "workspace sizing did not converge");
    }

private:
    int launch() {
        OVTK_HIP(hipSetDevice(device_));
        Workspace& ws = *ws_.ws;
        int e = 0;
        e = e ? e : ws.row_stage.ensure(size_t(n_rows_) * 4);
        e = e ? e : ws.row_cnt.ensure(size_t(n_rows_) * 4);
        e = e ? e : ws.row_used.ensure(size_t(n_rows_) * 4);
        e = e ? e : ws.stage.ensure(size_t(stage_cap_) * 4 + 64);   // (+ slack: compact_flat's 16-byte loads may read behind the last entry)
        e = e ? e : ws.deferred.ensure(size_t(shard_cap_) * kShards * sizeof(DeferredPiece));
        e = e ? e : ws.exact.ensure(size_t(exact_cap_) * sizeof(ExactPiece));
        e = e ? e : ws.scratch.ensure(size_t(scratch_cap_));
        e = e ? e : ws.wave_off.ensure(size_t(grid_ * kWavesPerBlock + 1) * sizeof(long long));
        e = e ? e : ws.tiles.ensure(size_t(n_tiles_ + 1) * sizeof(long long));
        const bool fold = fold_tail_ && n_rows_ <= kFoldTailRows;
        size_t status_bytes = sizeof(RunStatus) + (fold ? size_t(n_tiles_) * 4 + 16 : 0);  // + tile_cnt, zeroed with the status (+ slack: compact_kernel reads it 16 bytes at a time)
        // the short path (span_kernel.hpp): tile_cnt is summed by the lookup kernels' waves, every wave of compact_kernel sums the tiles in
        // front of its own
        const int sp_mode = short_path_mode().load(std::memory_order_relaxed);
        small_ = small_ && !(short_ok_ && sp_mode == 2);   // (mode 2: tests drive it with small batches)
        const bool span_sums = short_ok_ && !small_ && self_alloc_ && fold && sp_mode != 0 && !force_long_;
        const size_t status_stride = (status_bytes + 255) & ~size_t(255);   // two blocks: this call's, and the one compact_kernel zeroes for the next
        e = e ? e : ws.status.ensure(2 * status_stride);
        if (fold) e = e ? e : ws.gen[4].ensure(size_t(n_rows_) * 4);
        if (self_alloc_) e = e ? e : ws.gen[5].ensure(size_t(n_rows_) * 4);
        if (e) return e;
        EncodeWork w{};
        w.n_waves = grid_ * kWavesPerBlock;
        w.fold_tail = fold;
        uint8_t* const status_base = ws.status.as<uint8_t>();
        w.tile_cnt = fold ? reinterpret_cast<int32_t*>(status_base + sizeof(RunStatus)) : nullptr;   // (the one-launch path: block 0; the large path moves both below)
        w.row_emit = fold ? ws.gen[4].as<int32_t>() : nullptr;
        w.pending_rows = self_alloc_ ? ws.gen[5].as<int32_t>() : nullptr;
        w.out_cap = out_.data_capacity;
        w.rows_per_ticket = self_alloc_ ? row_tickets().load(std::memory_order_relaxed) : 0;
        w.wave_off = self_alloc_ ? nullptr : ws.wave_off.as<long long>();
        w.stage_region = int32_t(stage_cap_ / kShards);
        w.row_stage = ws.row_stage.as<int32_t>();
        w.row_cnt = ws.row_cnt.as<int32_t>();
        w.row_used = ws.row_used.as<int32_t>();
        w.tile_off = ws.tiles.as<long long>();
        w.stage = ws.stage.as<int32_t>();
        w.stage_cap = int32_t(stage_cap_);
        w.stage16 = stage16_ ? 1 : 0;
        w.deferred = ws.deferred.as<DeferredPiece>();
        w.shard_cap = int32_t(std::min<int64_t>(shard_cap_, INT32_MAX / kShards));
        w.exact = ws.exact.as<ExactPiece>();
        w.exact_cap = int32_t(exact_cap_);
        w.scratch = ws.scratch.as<uint8_t>();
        w.scratch_cap = uint32_t(std::min<int64_t>(scratch_cap_, 0xFFFFFFF0ll));
        w.status = ws.status.as<RunStatus>();
        phase_ = 0;

        if (small_ && fold) {  // one launch: blocks look their rows up, the last one merges, compacts and reports
            w.small = 1;
            const int blocks = std::max(1, std::min((n_rows_ + kWavesPerBlock - 1) / kWavesPerBlock, kSmallRows / kWavesPerBlock));
            w.n_waves = blocks * kWavesPerBlock;
            w.out_ids = d_ids_;
            w.out_begins = d_begins_;
            w.out_ends = d_ends_;
            w.host_status = ws.host_status;
            w.status_words = int32_t(status_bytes / 4);
            // the kernel leaves the device-side status zeroed behind itself: the memset is for a workspace it has not seen
            if (ws.clean_status != w.status || ws.clean_bytes < status_bytes || ws.clean_after_lease + 1 != ws.lease_count)
                OVTK_HIP(hipMemsetAsync(w.status, 0, status_bytes, s_));
            ws.clean_status = nullptr;   // (clean again once this call's kernel has run: finish())
            pending_clean_ = w.status;
            pending_clean_bytes_ = status_bytes;
            std::memset(ws.host_status, 0, sizeof(RunStatus));  // the kernel fills the scalar fields and shard_count[0] only
            ws.host_status->flags = kFlagDidNotRun;              // overwritten by the kernel; one that did not run leaves this (finish: OVTK_E_HIP)
            middle_(ws, d_in_, w, blocks);
            OVTK_HIP(hipEventRecord(ws.done, s_));
            return OVTK_OK;
        }
        ws.clean_status = nullptr;  // (the ordinary launches below leave their counters in the status block)
        // This call's status block: the one the workspace's last large call zeroed from its compact_kernel, if that was the lease
        // right before this one and nothing moved; else either block, after a memset.
        ws.status_half ^= 1;
        uint8_t* mine = status_base + size_t(ws.status_half) * status_stride;
        uint8_t* other = status_base + size_t(ws.status_half ^ 1) * status_stride;
        w.status = reinterpret_cast<RunStatus*>(mine);
        w.tile_cnt = fold ? reinterpret_cast<int32_t*>(mine + sizeof(RunStatus)) : nullptr;
        w.next_status = reinterpret_cast<RunStatus*>(other);
        // the large path, up to kTileSumTiles tiles: no ticket and no scan at the end of the middle's last kernel -- every wave of
        // compact_kernel sums the counts in front of its tile (beyond that the sums cost compact_kernel more than the tail cost the
        // middle: config 4's 2 048 tiles, compact 29.5 -> 32.7 us for merge_kernel 67 -> 62)
        w.tile_sums = fold && (n_tiles_ <= kTileSumTiles || span_sums) ? 1 : 0;
        w.host_status = ws.host_status;
        w.status_words = int32_t(status_bytes / 4);
        if (ws.zeroed_status != mine || ws.zeroed_bytes < status_bytes || ws.zeroed_after_lease + 1 != ws.lease_count)
            OVTK_HIP(hipMemsetAsync(w.status, 0, status_bytes, s_));
        ws.zeroed_status = nullptr;   // (until this call's compact_kernel is on its way)
        std::memset(ws.host_status, 0, sizeof(RunStatus));   // compact_kernel fills the scalar fields and the shards' counts
        ws.host_status->flags = kFlagDidNotRun;               // overwritten by the kernel; one that did not run leaves this (finish: OVTK_E_HIP)
        if (!self_alloc_)
            OVTK_LAUNCH(ws.marks, "prep_rows", prep_rows_kernel, std::min(grid_, kTicketBlocks), kBlockThreads, s_, d_in_, mul_, w);
        w.launch_mask = kLaunchAll;
        if (span_sums) {
            w.span_sums = 1;
            // (mode 2 leaves out everything it can, whatever the handle expects: the tests' way to the second set of launches)
            w.skip_mask = sp_mode == 2 ? (kSkipPending | kSkipMerge) : ((expect_pending_ ? 0 : kSkipPending) | (expect_merge_ ? 0 : kSkipMerge));
            w.launch_mask = kLaunchSpan | ((w.skip_mask & kSkipPending) ? 0 : kLaunchPending) | ((w.skip_mask & kSkipMerge) ? 0 : kLaunchMerge);
            w.merge_hint = int32_t(std::min<int64_t>(merge_hint_, INT32_MAX));
            w_ = w;
            phase_ = 1;
        }
        middle_(ws, d_in_, w, grid_);
        return launch_tail(ws, w, other, status_bytes);
    }

    // The short path, after all: a kernel of the middle that was left out had work.  The status block, the staging buffer, the row
    // records, the deferred list and the tile sums are as the kernels that did run left them; what is launched now adds to them.
    int launch_phase2(const RunStatus& st) {
        OVTK_HIP(hipSetDevice(device_));
        Workspace& ws = *ws_.ws;
        const bool pending_now = (w_.skip_mask & kSkipPending) && st.n_pending > 0;
        if (pending_now && !(w_.skip_mask & kSkipMerge)) {
            // merge_kernel has been through a list that the left-over rows' kernel is about to add to: from the start, every kernel
            force_long_ = true;
            return launch();
        }
        EncodeWork w = w_;
        w.merge_hint = st.n_unresolved;   // (counted by the kernels that have run; the left-over rows' kernel may add to it)
        w.launch_mask = (pending_now ? kLaunchPending : 0) | ((w_.skip_mask & kSkipMerge) ? kLaunchMerge : 0);
        w.skip_mask = pending_now || !(w_.skip_mask & kSkipPending) ? 0 : kSkipPending;   // (rows were left over or they were not: nothing is left out that had work)
        // what the first set's last kernels left in the status block: the row width and the capacity verdict of row_width_kernel /
        // compact_kernel (made from row records that lacked rows or pieces) and the ticket they drew
        static_assert(offsetof(RunStatus, width_ticket) == offsetof(RunStatus, width) + 4, "width and width_ticket are cleared together");
        OVTK_HIP(hipMemsetAsync(&w.status->width, 0, 8, s_));
        OVTK_HIP(hipMemsetAsync(&w.status->flags, 0, 4, s_));   // (no other flag is set: finish() came here past every one of them)
        OVTK_HIP(hipMemsetAsync(&w.status->n_out, 0, 4, s_));
        std::memset(ws.host_status, 0, sizeof(RunStatus));
        ws.host_status->flags = kFlagDidNotRun;
        middle_(ws, d_in_, w, grid_);
        phase_ = 2;
        return launch_tail(ws, w, zeroed_other_, zeroed_other_bytes_);
    }

    // count_scan (where the middle's last kernel does not fold it) and compact_kernel, the call's last kernel.
    int launch_tail(Workspace& ws, EncodeWork& w, const void* other, size_t status_bytes) {
        if (!w.fold_tail)
            OVTK_LAUNCH(ws.marks, "count_scan", count_scan_kernel, std::min((n_tiles_ + 3) / 4, kTicketBlocks), kBlockThreads, s_,
                        n_rows_, w, (long long)out_.data_capacity);
        // (compact_kernel's waves take work items of kCompactRows rows: a wave per item where the chip holds that many -- the kernel is a chain
        // of memory round trips per item, not a matter of instructions)
        // Few, long rows (8 KB: an item is ~7 000 ids): several waves share an item -- where the item has no unused entry its copy is dealt out
        // among them in steps of 1 024 ids.  (A row of n bytes has at most n ids, ~n / 4.5 on text: the split follows the bytes.)
        int split = 1;
        if (!dense_on_ && !wire_.hdr) {
            const long long ids_per_item = (long long)kCompactRows * (in_.strings.n_chars / std::max<long long>(1, n_rows_)) / 4;
            while (split < 16 && ids_per_item >= 2048ll * split && (long long)(n_rows_ / kCompactRows) * split * 2 <= std::max<long long>((long long)device_cu_count(device_) * 64, 1024)) split *= 2;
        }
        w.compact_split = split;
        const long long units = ((long long)n_rows_ + kCompactRows - 1) / kCompactRows * split;
        const int cgrid = int(std::max<long long>(1, std::min<long long>((units + kWavesPerBlock - 1) / kWavesPerBlock, (long long)device_cu_count(device_) * 16)));
        const RaggedSink rsink{d_ids_, d_begins_, d_ends_};
        if (dense_on_) {
            OVTK_LAUNCH(ws.marks, "row_width", row_width_kernel, std::min(64, (n_rows_ + kBlockThreads - 1) / kBlockThreads), kBlockThreads, s_, n_rows_, w, dense_);
            if (stage16_) OVTK_LAUNCH(ws.marks, "compact", (compact_kernel<DenseSink, true>), cgrid, kBlockThreads, s_, n_rows_, w, dense_);
            else OVTK_LAUNCH(ws.marks, "compact", (compact_kernel<DenseSink, false>), cgrid, kBlockThreads, s_, n_rows_, w, dense_);
        } else if (wire_.hdr && stage16_) OVTK_LAUNCH(ws.marks, "compact", (compact_kernel<WireSink, true>), cgrid, kBlockThreads, s_, n_rows_, w, wire_);
        else if (wire_.hdr) OVTK_LAUNCH(ws.marks, "compact", (compact_kernel<WireSink, false>), cgrid, kBlockThreads, s_, n_rows_, w, wire_);
        else if (stage16_) OVTK_LAUNCH(ws.marks, "compact", (compact_kernel<RaggedSink, true>), cgrid, kBlockThreads, s_, n_rows_, w, rsink);
        else OVTK_LAUNCH(ws.marks, "compact", (compact_kernel<RaggedSink, false>), cgrid, kBlockThreads, s_, n_rows_, w, rsink);
        // (the other status block is clean for the next lease of this workspace once this call's compact_kernel has RUN: noted in
        // finish(), after the event -- ADVICE r04: noted here, a run that was dropped without finish(), or whose launch failed
        // afterwards, left the next lease without its memset)
        pending_zeroed_ = other;
        pending_zeroed_bytes_ = status_bytes;
        zeroed_other_ = other;
        zeroed_other_bytes_ = status_bytes;
        OVTK_HIP(hipEventRecord(ws.done, s_));
        return OVTK_OK;
    }

    int device_;
    std::string op_;
    ovtk_ragged_strings in_;
    const uint8_t* skips_;
    int mul_;
    ovtk_ragged_i32_out out_;
    int mem_, in_mem_;
    bool direct_out_ = false;      // host-memory call whose output buffers are pinned: the kernels write them (no D2H copy)
    std::shared_ptr<void> keep_;
    std::vector<std::shared_ptr<WorkspaceLease>> others_;
    std::function<int()> front_check_;
    hipStream_t s_;
    Middle middle_;
    bool self_alloc_;
    int blocks_per_cu_;
    WorkspaceLease ws_;
    bool fold_tail_;  // the middle's last kernel finishes the row scan itself (BPE: merge_kernel) while the batch is small
    bool small_ok_ = false, small_ = false;
    bool stage16_ = false, stage_twice_ = false;
    bool short_ok_ = false, expect_pending_ = false, expect_merge_ = true, force_long_ = false;
    int64_t merge_hint_ = -1;
    int phase_ = 0;          // 1: the attempt in flight is the short path's first set of launches; 2: what it had left out was launched after all
    EncodeWork w_{};         // the attempt's work description (phase 2 launches with it)
    const void* zeroed_other_ = nullptr;
    size_t zeroed_other_bytes_ = 0;
    const void* pending_zeroed_ = nullptr;
    size_t pending_zeroed_bytes_ = 0;
    const void* pending_clean_ = nullptr;
    size_t pending_clean_bytes_ = 0;
    WireSink wire_{};
    DenseSink dense_{};
    bool dense_on_ = false;
    std::function<void(const RunStatus&)> on_status_;
    RowsIn d_in_{};
    int n_rows_ = 0, grid_ = 0, n_tiles_ = 0;
    int64_t stage_cap_ = 0, shard_cap_ = 0, exact_cap_ = 0, scratch_cap_ = 0;
    int32_t *d_begins_ = nullptr, *d_ends_ = nullptr, *d_ids_ = nullptr;
};

template <class Middle>
std::unique_ptr<RowsRun<std::decay_t<Middle>>> make_rows_run(int device, const char* op, const ovtk_ragged_strings* in,
                                                             const uint8_t* skips, int mul, const ovtk_ragged_i32_out* out, int mem,
                                                             hipStream_t s, Middle&& middle, bool self_alloc = false,
                                                             int blocks_per_cu = 6, bool tail_in_middle = false) {
    return std::make_unique<RowsRun<std::decay_t<Middle>>>(device, op, in, skips, mul, out, mem, s, std::forward<Middle>(middle),
                                                           self_alloc, blocks_per_cu, tail_in_middle);
}

// The synchronous form every *_run entry point uses.  `middle` is copied: capture by value.
template <class Middle>
int run_rows_to_ids(int device, const char* op, const ovtk_ragged_strings* in, const uint8_t* skips, int mul,
                    ovtk_ragged_i32_out* out, int mem, hipStream_t s, Middle&& middle, bool self_alloc = false,
                    int blocks_per_cu = 6, bool tail_in_middle = false) {
    auto run = make_rows_run(device, op, in, skips, mul, out, mem, s, std::forward<Middle>(middle), self_alloc, blocks_per_cu,
                             tail_in_middle);
    if (int rc = run->start()) return rc;
    return run->finish(out);
}

}  // namespace ovtk

// api_common.hpp -- helpers shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "encode_kernels.hpp"
#include "runtime.hpp"
#include "tables.hpp"

// Handle of RegexSplit (shared by api_encode.cpp and the fused WordPiece path in api_ops.cpp).
struct ovtk_regex_split {
    int device = 0;
    ovtk::SplitDev dev{};
    int mode = 1;  // 0 removed, 1 isolated, 2 merged-with-previous, 3 merged-with-next
    bool invert = false;
    int max_splits = -1;
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

// The "ragged strings in -> ragged i32 out" pipeline shared by BPETokenizer, the fused encode and
// WordpieceTokenizer: prep (validation + per-wave staging arenas) -> middle(ws, d_in, w, grid) (the op's kernels: ids
// into staging, per-row counts; the first one must be launched with `grid` blocks, the geometry prep summed over) ->
// count_scan (final offsets) -> compact.  Workspace overflows reported
// by the kernels are handled by growing the buffer and running again.
// self_alloc: the middle's first kernel (lookup_kernel) validates the offsets and takes its staging from bump allocators
// itself, so no prep launch is needed.
template <class Middle>
int run_rows_to_ids(int device, const char* op, const ovtk_ragged_strings* in, const uint8_t* skips, int mul,
                    ovtk_ragged_i32_out* out, int mem, hipStream_t s, Middle&& middle, bool self_alloc = false,
                    int blocks_per_cu = 6) {
    WorkspaceLease ws(device);
    if (!ws->host_status) return set_error(OVTK_E_HIP, "pinned host allocation failed");
    RowsIn d_in{};
    if (int rc = stage_input(*ws.ws, in, skips, mem, s, d_in)) return rc;

    const int n_rows = d_in.n_rows;
    int64_t stage_cap = std::min<int64_t>((in->strings.n_chars + in->strings.n) * mul, INT32_MAX - 1);
    int64_t shard_cap = std::max<int64_t>({4096, (in->strings.n_chars / 16 + in->strings.n / 4) / kShards,
                                           int64_t(ws->deferred.size() / sizeof(DeferredPiece) / kShards)});
    int64_t exact_cap = std::max<int64_t>(4096, int64_t(ws->exact.size() / sizeof(ExactPiece)));
    int64_t scratch_cap = std::max<int64_t>(ws->scratch.size(), int64_t(16) << 20);

    int32_t *d_begins = nullptr, *d_ends = nullptr, *d_ids = nullptr;
    if (int rc = out_target(ws->out_a, out->begins, size_t(n_rows) * 4, mem, &d_begins)) return rc;
    if (int rc = out_target(ws->out_b, out->ends, size_t(n_rows) * 4, mem, &d_ends)) return rc;
    if (int rc = out_target(ws->out_c, out->data, size_t(out->data_capacity) * 4, mem, &d_ids)) return rc;

    const int grid = grid_rows(device, n_rows, blocks_per_cu);
    const int n_tiles = (n_rows + kRowTile - 1) / kRowTile;
    if (self_alloc)  // every wave may leave one chunk partly unused
        stage_cap = std::min<int64_t>(stage_cap + int64_t(grid * kWavesPerBlock + kShards) * kStageChunk, INT32_MAX - 1);
    for (int attempt = 0; attempt < 6; ++attempt) {
        int e = 0;
        e = e ? e : ws->row_stage.ensure(size_t(n_rows) * 4);
        e = e ? e : ws->row_cnt.ensure(size_t(n_rows) * 4);
        e = e ? e : ws->row_used.ensure(size_t(n_rows) * 4);
        e = e ? e : ws->stage.ensure(size_t(stage_cap) * 4);
        e = e ? e : ws->deferred.ensure(size_t(shard_cap) * kShards * sizeof(DeferredPiece));
        e = e ? e : ws->exact.ensure(size_t(exact_cap) * sizeof(ExactPiece));
        e = e ? e : ws->scratch.ensure(size_t(scratch_cap));
        e = e ? e : ws->wave_off.ensure(size_t(grid * kWavesPerBlock + 1) * sizeof(long long));
        e = e ? e : ws->tiles.ensure(size_t(n_tiles + 1) * sizeof(long long));
        e = e ? e : ws->status.ensure(sizeof(RunStatus));
        if (e) return e;
        EncodeWork w{};
        w.n_waves = grid * kWavesPerBlock;
        w.wave_off = self_alloc ? nullptr : ws->wave_off.as<long long>();
        w.stage_region = int32_t(stage_cap / kShards);
        w.row_stage = ws->row_stage.as<int32_t>();
        w.row_cnt = ws->row_cnt.as<int32_t>();
        w.row_used = ws->row_used.as<int32_t>();
        w.tile_off = ws->tiles.as<long long>();
        w.stage = ws->stage.as<int32_t>();
        w.stage_cap = int32_t(stage_cap);
        w.deferred = ws->deferred.as<DeferredPiece>();
        w.shard_cap = int32_t(std::min<int64_t>(shard_cap, INT32_MAX / kShards));
        w.exact = ws->exact.as<ExactPiece>();
        w.exact_cap = int32_t(exact_cap);
        w.scratch = ws->scratch.as<uint8_t>();
        w.scratch_cap = uint32_t(std::min<int64_t>(scratch_cap, 0xFFFFFFF0ll));
        w.status = ws->status.as<RunStatus>();

        OVTK_HIP(hipMemsetAsync(w.status, 0, sizeof(RunStatus), s));
        if (!self_alloc)
            OVTK_LAUNCH(ws->marks, "prep_rows", prep_rows_kernel, std::min(grid, kTicketBlocks), kBlockThreads, s, d_in, mul, w);
        middle(*ws.ws, d_in, w, grid);
        OVTK_LAUNCH(ws->marks, "count_scan", count_scan_kernel, std::min((n_tiles + 3) / 4, kTicketBlocks), kBlockThreads, s,
                    n_rows, w, (long long)out->data_capacity);
        OVTK_LAUNCH(ws->marks, "compact", compact_kernel, grid_lookup(device, n_rows), kBlockThreads, s, n_rows, w, d_ids, d_begins, d_ends);
        if (int rc = finish_status(*ws.ws, s)) return rc;

        const RunStatus& st = *ws->host_status;
        if (st.flags & kFlagRange) return set_error(OVTK_E_RANGE, "input begins/ends index outside their tensors");
        if (st.flags & kFlagStageOverflow) {
            const int64_t need = self_alloc ? stage_cap * 2 : int64_t(st.stage_need);  // the allocators do not know the total
            if (need >= INT32_MAX - 1)
                return set_error(OVTK_E_UNSUPPORTED, "batch needs more than 2^31 staging entries; split the call");
            stage_cap = need;
            continue;
        }
        if (st.flags & kFlagDeferOverflow) {
            int64_t most = 0;
            for (int k = 0; k < kShards; ++k) most = std::max<int64_t>(most, st.shard_count[k * kCounterStride]);
            shard_cap = most + most / 8 + 256;
            continue;
        }
        if (st.flags & kFlagExactOverflow) {
            exact_cap = int64_t(st.n_exact) + 256;
            continue;
        }
        if (st.flags & kFlagScratchOverflow) {
            scratch_cap = std::max<int64_t>(scratch_cap * 2, int64_t(st.scratch_used) + (1 << 20));
            if (scratch_cap > (int64_t(3) << 30))
                return set_error(OVTK_E_UNSUPPORTED, "exact-path scratch would exceed 3 GiB; split the call");
            continue;
        }
        if (st.flags & kFlagOutCapacity)
            return set_error(OVTK_E_CAPACITY, std::string(op) + ": output ids buffer too small (" +
                                                  std::to_string(st.n_out) + " ids, capacity " +
                                                  std::to_string(out->data_capacity) + ")");
        out->n_data = st.n_out;
        if (int rc = copy_back(out->begins, d_begins, size_t(n_rows) * 4, mem, s)) return rc;
        if (int rc = copy_back(out->ends, d_ends, size_t(n_rows) * 4, mem, s)) return rc;
        if (int rc = copy_back(out->data, d_ids, size_t(st.n_out) * 4, mem, s)) return rc;
        if (mem == OVTK_MEM_HOST) OVTK_HIP(hipStreamSynchronize(s));
        return OVTK_OK;
    }
    return set_error(OVTK_E_HIP, "workspace sizing did not converge");
}

}  // namespace ovtk

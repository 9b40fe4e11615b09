// api_common.hpp -- helpers shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "encode_kernels.hpp"
#include "runtime.hpp"
#include "tables.hpp"

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

}  // namespace ovtk

// rr_common.hpp -- includes, error helper, constants and index helpers shared by the kernels and the executor.
// Part of the one translation unit rr_engine.hip builds (included from there first).
#pragma once

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/rr_hip.h"
#include "rr_plan.hpp"

#define RR_VERSION_NUM 100

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(RR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                 \
    } while (0)

constexpr int kBlock = 256;
constexpr int kSampleGroup = 16;   // routing-tick launches per HIP-event bracket
// lag[] carries two flag bits for the boundary reaches of a partitioned network
constexpr int32_t kGhostBit = 1 << 30;    // value prescribed from the ghost series (an upstream reach owned by another GPU)
constexpr int32_t kExportBit = 1 << 29;   // value also copied to the export series (feeds another GPU)
constexpr int32_t kTileGhostBit = rr::kTileGhost;     // tile layout only: position mirrors a reach another tile owns
constexpr int32_t kTileExportBit = rr::kTileExport;   // tile layout only: reach is mirrored by a ghost, values also go to the export ring
constexpr int32_t kLagMask = kTileExportBit - 1;

// Index arithmetic: every tick, row and chunk index of a call is below 2^31 (checked in session_begin), and the
// hardware has no integer divide -- a 64-bit `%` is a ~100-instruction emulation.  Division by a run-time constant
// goes through a host-computed double reciprocal instead: floor(x * (1/d)) is exact or one too small (only when d
// divides x), which the single fix-up repairs.
struct Div32 {
    uint32_t d;
    double inv;
    Div32() = default;
    __host__ __device__ explicit Div32(uint32_t d_) : d(d_ ? d_ : 1u), inv(1.0 / (double)(d_ ? d_ : 1u)) {}
    __device__ __forceinline__ uint32_t div(uint32_t x, uint32_t &rem) const
    {
        uint32_t q = (uint32_t)((double)x * inv);
        uint32_t r = x - q * d;
        if (r >= d) { r -= d; ++q; }
        rem = r;
        return q;
    }
    __device__ __forceinline__ uint32_t mod(uint32_t x) const { uint32_t r; div(x, r); return r; }
};

// Row addressing of a (rows, ld) array read or written cyclically: row of step t is (t - t0) % rows.
struct RowView {
    double *base;
    int64_t ld;
    int64_t t0;
    Div32 rows;
    RowView() = default;
    RowView(double *base_, int64_t ld_, int64_t t0_, uint32_t rows_) : base(base_), ld(ld_), t0(t0_), rows(rows_) {}
    __device__ __forceinline__ int64_t offset(int64_t t) const { return (int64_t)rows.mod((uint32_t)(t - t0)) * ld; }      // in elements
    __device__ __forceinline__ double *row(int64_t t) const { return base + offset(t); }
};

// v_perm_b32 selectors: the four bytes of a word as they are / reversed (a NetCDF-3 file stores big-endian values)
constexpr uint32_t kSelNative = 0x03020100u, kSelSwap = 0x00010203u;
__device__ __forceinline__ float f32_from_file(float raw, uint32_t sel)
{
    uint32_t bits;
    __builtin_memcpy(&bits, &raw, 4);
    bits = __builtin_amdgcn_perm(bits, bits, sel);
    float v;
    __builtin_memcpy(&v, &bits, 4);
    return v;
}
__device__ __forceinline__ float f32_to_file(float v, uint32_t sel) { return f32_from_file(v, sel); }

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// Workgroups are handed to the eight XCDs in turn (blockIdx % 8 labels the workgroups that share an XCD and its L2, cdna
// programming guide section 5.5 T1).  Kernels whose neighbouring tiles touch the same cache lines -- column tiles of rows
// whose pitch is not a multiple of a line -- take tile xcd_swizzle(blockIdx) instead of blockIdx, so that neighbours run on
// one XCD at about the same time: a line two tiles share is fetched once and written back whole.  Bijective for any grid.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t nwg)
{
    const uint32_t q = nwg >> 3, r = nwg & 7u, x = b & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (b >> 3);
}

}  // namespace

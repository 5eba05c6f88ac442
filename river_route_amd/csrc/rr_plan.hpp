// Host-side network analysis for the routing engine: turns the reference's CSC adjacency
// (river_route/tools.py:75-109) into the "lag-ordered" layout the HIP kernels stream over.
//
// Layout idea: every reach has exactly one downstream reach, so the distance D(i) to
// its outlet satisfies D(up) = D(down) + 1 on EVERY edge.  Reaches are placed farthest-from-outlet first,
// level by level, each level in the order of its downstream reaches.  Then
//   * lag(p) = Dmax - D is non-decreasing in the engine position p;
//   * the reaches flowing into position p occupy the contiguous position range
//     [child_ptr[p], child_ptr[p+1]) and child_ptr is simply the running count of upstream reaches,
//     so the adjacency needs no column-index array at all.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rr {

struct HostPlan {
    int64_t n = 0;
    int64_t n_edges = 0;
    int32_t depth = 0;          // number of levels = Dmax + 1
    int64_t widest_level = 0;
    int64_t n_headwaters = 0;
    int64_t n_outlets = 0;
    bool identity = false;      // perm[p] == p for all p

    std::vector<int32_t> perm;       // [n]   params index at engine position p
    std::vector<int32_t> inv;        // [n]   engine position of params index i
    std::vector<int32_t> lag;        // [n]   Dmax - D(perm[p]); non-decreasing in p
    std::vector<int32_t> child_ptr;  // [n+1] prefix count of upstream reaches in engine order
    std::vector<uint16_t> hw_children;  // [n] how many of p's upstream reaches are headwaters (they come first)
    std::vector<int64_t> lag_start;  // [depth+1] first engine position with lag >= d
    std::vector<int32_t> edge_of;    // [n]   CSC entry index of the edge leaving params index i, -1 at outlets
    std::vector<int32_t> inner_pos;  // [n_inner] engine position of the k-th inner reach (ascending params order)
};

// Two-phase tiled permutation dst[q] = src[pi[q]].  A random 8-byte gather wastes 7/8
// of every 64-byte sector it touches, and with eight private L2s no single pass can avoid that; instead
//   phase A: each block reads one SOURCE tile coalesced, sorts it in LDS by destination tile and appends the
//            pieces to an intermediate row M (runs of elements bound for the same destination tile);
//   phase B: each block reads one DESTINATION tile's bucket of M (contiguous), places it in LDS and writes
//            the destination tile coalesced.
// Every global access of both phases is a contiguous run; LDS does the shuffling.
struct TiledPermutation {
    int64_t n = 0;
    int32_t tile = 0;                // elements per tile
    std::vector<uint16_t> slot_a;    // [n] LDS slot of source element i inside its tile (sorted by destination)
    std::vector<int32_t> m_index;    // [n] for source tile a, sorted slot m: index in M  (stored at a*tile + m)
    std::vector<uint16_t> slot_b;    // [n] for M index g: offset of its destination inside the destination tile
};
void build_tiled_permutation(const int32_t *pi, int64_t n, int32_t tile, TiledPermutation &out);

// Cuts a forest (down[i] = downstream reach or -1, upstream reaches have smaller indices) into at most n_parts
// parts so that the largest part is as small as the greedy allows and the part graph stays acyclic: every
// piece between cuts is connected, and only pieces the same number of cuts away from their outlet share a part.
// Parts are numbered upstream-first.  part_of[n], sizes[parts].
void partition_forest(const std::vector<int32_t> &down, int32_t n_parts, int32_t *part_of, std::vector<int64_t> &sizes);

// Returns 0 or an RR_E_* code with a message in err.
int build_host_plan(int64_t n, const int32_t *indptr, const int32_t *indices, HostPlan &plan, std::string &err);

}  // namespace rr

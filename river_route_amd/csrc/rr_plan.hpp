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
    std::vector<int32_t> down;       // [n]   downstream params index, -1 at outlets
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


// ---- subtree tiles: the layout of the time-tiled kernel ----
//
// The time-tiled kernel advances a TILE of at most `block` positions by K routing ticks per launch with everything
// but the lateral/discharge records resident on chip.  A tile is a set of reaches closed under "upstream within the
// tile": every upstream reach of a tile reach is either in the tile or is mirrored in it by a GHOST position whose
// record the tile that owns the reach writes (one 8-byte store per tick into the ghost's record slot).  Tiles form a
// DAG; a tile of level l runs chunk c in launch l + c, so the schedule's skew is (levels x K) ticks and no two tasks
// of one launch depend on each other.
//   * SMALL subtrees (at most `block` reaches, hanging off a reach with more than that upstream) are complete:
//     they need no ghost, have level 0, and are bin-packed into tiles, neighbours in lag first (a tile is busy from
//     its smallest to its largest lag, so tiles of similar lag waste the fewest ticks while the pipeline fills);
//   * the remaining reaches (the SKELETON: ~1/sqrt(pi block) of a random network) are cut bottom-up into connected
//     pieces of at most `block` positions, ghosts included; a piece's level is one more than the deepest piece or
//     small subtree that feeds it, and pieces of one level share tiles.
// Positions of a tile are in breadth-first order from the tile's outlets, so the upstream positions of a position
// are contiguous: [cfirst[p], cfirst[p] + count), headwater tributaries first (UnitMuskingum needs them apart).
constexpr int32_t kTileGhost = 1 << 28;    // lag[] flag: position mirrors a reach owned by another tile
constexpr int32_t kTileExport = 1 << 27;   // lag[] flag: reach is mirrored by the ghost at position xpos[p]
struct TilePlan {
    bool ok = false;                 // false: a reach has more upstream reaches than a tile holds (use the streaming kernel)
    int32_t block = 0;               // capacity of a tile in positions
    int64_t np = 0;                  // positions = reaches + ghosts
    int64_t n_ghost = 0;
    int32_t n_tiles = 0, n_levels = 0;
    std::vector<int32_t> tile_ptr;     // [n_tiles + 1] first position of each tile
    std::vector<int32_t> tile_level;   // [n_tiles] non-decreasing
    std::vector<int32_t> tile_lag_lo, tile_lag_hi;   // [n_tiles] smallest / largest lag of the tile's positions
    std::vector<int32_t> tile_flags;   // [n_tiles] 1: a position of the tile has more than three upstream positions (not for k_tile's short tick)
    std::vector<int32_t> level_start;  // [n_levels + 1] first tile of each level
    std::vector<int32_t> perm;         // [np] params index of the reach at (or mirrored by) a position
    std::vector<int32_t> inv;          // [n]  position of reach i
    std::vector<int32_t> lag;          // [np] lag | kTileGhost | kTileExport
    std::vector<int32_t> cfirst;       // [np] first upstream position
    std::vector<uint32_t> ccnt;        // [np] upstream positions | headwater tributaries among them << 16
    std::vector<int32_t> xpos;         // [np] kTileExport: position of the mirroring ghost; kTileGhost: position of the mirrored reach; else -1
    std::vector<int32_t> tile_of;      // [np]
    std::vector<int32_t> ext_ghost;    // skeleton-only plans: [n] position of the ghost that mirrors reach i where i is NOT in the plan, else -1
};
// down[i]: downstream reach or -1 (upstream reaches have smaller indices); lag_of[i]: levels between reach i and the
// farthest headwater of the network.
// big (optional, [n]): which reaches form the skeleton (default: more than `block` reaches upstream).  skeleton_only: only those
// get positions; every other reach that flows into one is mirrored by a ghost whose record somebody else writes
// (TilePlan::ext_ghost), tile levels start at 1.
void build_tile_plan(const std::vector<int32_t> &down, const std::vector<int32_t> &lag_of, int32_t block, TilePlan &out,
                     const std::vector<uint8_t> *big = nullptr, bool skeleton_only = false);

// ---- direct tiles: rows in params order read and written by the routing kernel itself ----
//
// Where the params order numbers every small subtree contiguously (any depth-first post-order does: a subtree is then a
// column range ending at its outlet), a tile is a RANGE of columns [c0, c0 + nc): lane = column.  The tile's lanes are whole
// small subtrees -- at most `lanes` reaches and at most `wmax` levels high, so that the lags of a tile span less than wmax --
// and the columns of skeleton reaches that lie between them (HOLES: a reach with a large or tall subtree comes right after its
// last tributary's subtree).  A tile has no ghost and no level: a lane's upstream lanes are in the tile.  The kernel reads a
// row segment per tick, holds it in an LDS window until the lane whose turn it is has routed it (delay = lag - smallest lag of
// the tile), and writes finished row segments back: no record ring, no permutation pass for these columns.  The skeleton
// (TilePlan, skeleton_only) keeps records: a hole lane forwards its column's lateral inflow to the skeleton position's record,
// the outlet of a small subtree sends its discharge to the ghost that mirrors it there, and a small pass patches the holes of
// the output rows from the skeleton's records.
constexpr int32_t kDirectHole = 1 << 30;      // delay[] flag: the column belongs to the skeleton
constexpr int32_t kDirectSenderShift = 8, kDirectDelayMask = 0xFF;      // delay[] bits 8-14: 1 + the column's number among its tile's senders (0: none); bits 0-7: the delay
constexpr int32_t kDirectExport = 1 << 15;    // delay[] flag: a boundary export of a partitioned network routed by a lane; xinfo = its column in the export series
struct DirectPlan {
    bool ok = false;
    std::string why;                   // not ok: the first reason
    int32_t lanes = 0, wmax = 0;
    int32_t n_tiles = 0;
    int64_t n_holes = 0, n_exports = 0;
    std::vector<int32_t> tile_c0, tile_nc, tile_lag_lo, tile_span;   // [n_tiles]; span = largest - smallest lag of the tile's lanes
    std::vector<int32_t> delay;        // [n] lag - tile_lag_lo of the column's tile (bits 0-7) | sender number + 1 (bits 8-14) | kDirectHole
    std::vector<int32_t> up3;          // [n] three 10-bit lane numbers of the upstream reaches (0x3FF: none), headwaters first; bits 30, 31: how many are headwaters
    std::vector<int32_t> xinfo;        // [n] hole: position of the reach in `skel`; outlet of a small subtree below a skeleton reach: position of its ghost there; else -1
    std::vector<uint8_t> big;          // [n] 1: skeleton
    std::vector<int32_t> send_ptr;     // [n_tiles + 1] the tile's senders (holes and outlets that feed the skeleton, at most kDirectSenders), in column order
    std::vector<int32_t> send_lane;    // [senders] lane of the tile | kDirectHole
    TilePlan skel;                     // the skeleton's tiles (levels from 1)
};
constexpr int32_t kDirectSenders = 64;      // one wave forwards them
// Boundary reaches of a partitioned network (optional, both [n]): ghost[i] != 0 -- the reach's discharge is prescribed (it is routed on
// another GPU): its column is passed through like a hole's and nobody sends for it (the boundary series is turned into the record of the
// skeleton's ghost that mirrors it by the in-pass); it must be a headwater of this network, and the reaches downstream of it join the skeleton.
// export_slot[i] >= 0 -- the reach's discharge also goes to that column of the export series: a skeleton reach's by the skeleton's
// kernel, a lane's by the lane (kDirectExport); it must be an outlet of this network.
void build_direct_plan(const std::vector<int32_t> &down, const std::vector<int32_t> &lag_of, int32_t lanes, int32_t wmax, int32_t skel_block,
                       DirectPlan &out, const std::vector<uint8_t> *ghost = nullptr, const std::vector<int32_t> *export_slot = nullptr);

// Depth-first post-order of a forest given by downstream indices in any order (include/rr_hip.h: rr_postorder).  False: an index out of
// range or a cycle.
bool postorder(const int64_t *down, int64_t n, int64_t *order);

// Returns 0 or an RR_E_* code with a message in err.
int build_host_plan(int64_t n, const int32_t *indptr, const int32_t *indices, HostPlan &plan, std::string &err);

}  // namespace rr

#include "rr_plan.hpp"

#include <algorithm>
#include <cstdlib>
#include <limits>
#include <utility>
#include <vector>

#include "../../include/rr_hip.h"

namespace rr {

void build_tiled_permutation(const int32_t *pi, int64_t n, int32_t tile, TiledPermutation &T)
{
    T = TiledPermutation();
    T.n = n;
    T.tile = tile;
    if (n == 0) return;
    const int64_t nt = (n + tile - 1) / tile;
    std::vector<int32_t> q_of(n);
    for (int64_t q = 0; q < n; ++q) q_of[pi[q]] = (int32_t)q;
    // run (b, a): elements of source tile a bound for destination tile b; M is ordered by (b, a, source index)
    std::vector<int64_t> run(nt * nt + 1, 0);
    for (int64_t i = 0; i < n; ++i) ++run[(int64_t)(q_of[i] / tile) * nt + i / tile + 1];
    for (size_t k = 1; k < run.size(); ++k) run[k] += run[k - 1];
    std::vector<int32_t> g_of(n);
    for (int64_t i = 0; i < n; ++i) g_of[i] = (int32_t)run[(int64_t)(q_of[i] / tile) * nt + i / tile]++;
    T.slot_a.resize(n); T.m_index.resize(n); T.slot_b.resize(n);
    std::vector<std::pair<int32_t, int32_t>> tmp;
    for (int64_t a = 0; a < nt; ++a) {
        const int64_t i0 = a * tile, i1 = std::min<int64_t>(n, i0 + tile);
        tmp.clear();
        for (int64_t i = i0; i < i1; ++i) tmp.emplace_back(g_of[i], (int32_t)i);
        std::sort(tmp.begin(), tmp.end());
        for (size_t m = 0; m < tmp.size(); ++m) {
            T.slot_a[tmp[m].second] = (uint16_t)m;
            T.m_index[i0 + (int64_t)m] = tmp[m].first;
        }
    }
    for (int64_t i = 0; i < n; ++i) T.slot_b[g_of[i]] = (uint16_t)(q_of[i] % tile);
}

namespace {

struct Pieces {
    std::vector<uint8_t> cut;        // edge below reach i is cut
    std::vector<int32_t> root_of;    // piece root (reach whose downstream edge is cut, or an outlet)
    std::vector<int32_t> roots;      // all piece roots
    std::vector<int64_t> size;       // by root reach index
    std::vector<int32_t> depth;      // cuts between the piece and its outlet, by root reach index
};

// Bottom-up greedy: a reach keeps its upstream residuals while they fit under `cap`, cutting the heaviest first.
void cut_forest(const std::vector<int32_t> &down, const std::vector<int32_t> &up_ptr, const std::vector<int32_t> &up_idx,
                int64_t cap, Pieces &P)
{
    const int64_t n = (int64_t)down.size();
    P.cut.assign(n, 0);
    std::vector<int64_t> res(n, 0);
    std::vector<std::pair<int64_t, int32_t>> kids;
    for (int64_t v = 0; v < n; ++v) {
        int64_t total = 1;
        kids.clear();
        for (int32_t e = up_ptr[v]; e < up_ptr[v + 1]; ++e) { kids.emplace_back(res[up_idx[e]], up_idx[e]); total += res[up_idx[e]]; }
        if (total > cap) {
            std::sort(kids.begin(), kids.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            for (const auto &k : kids) {
                if (total <= cap) break;
                P.cut[k.second] = 1;
                total -= k.first;
            }
        }
        res[v] = total;
    }
    P.root_of.assign(n, -1); P.size.assign(n, 0); P.depth.assign(n, 0); P.roots.clear();
    for (int64_t v = n - 1; v >= 0; --v) {
        if (down[v] < 0 || P.cut[v]) {
            P.root_of[v] = (int32_t)v;
            P.depth[v] = down[v] < 0 ? 0 : P.depth[P.root_of[down[v]]] + 1;
            P.roots.push_back((int32_t)v);
        } else {
            P.root_of[v] = P.root_of[down[v]];
        }
        ++P.size[P.root_of[v]];
    }
}

// First-fit-decreasing packing of the pieces of each depth class into bins of capacity cap.
// Returns the number of bins; bin_of[root] receives the bin.
int64_t pack_pieces(const Pieces &P, int64_t cap, std::vector<int32_t> &bin_of, std::vector<int64_t> &bin_size,
                    std::vector<int32_t> &bin_depth)
{
    std::vector<int32_t> order(P.roots);
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (P.depth[a] != P.depth[b]) return P.depth[a] > P.depth[b];
        if (P.size[a] != P.size[b]) return P.size[a] > P.size[b];
        return a < b;
    });
    bin_of.assign(P.size.size(), -1);
    bin_size.clear(); bin_depth.clear();
    size_t class_begin = 0;
    int32_t cur_depth = -1;
    for (int32_t r : order) {
        if (P.depth[r] != cur_depth) { cur_depth = P.depth[r]; class_begin = bin_size.size(); }
        size_t b = class_begin;
        for (; b < bin_size.size(); ++b) if (bin_size[b] + P.size[r] <= cap) break;
        if (b == bin_size.size()) { bin_size.push_back(0); bin_depth.push_back(cur_depth); }
        bin_size[b] += P.size[r];
        bin_of[r] = (int32_t)b;
    }
    return (int64_t)bin_size.size();
}

}  // namespace

// "Trunk" partition: every maximal subtree of at most n / (4 parts) reaches is a piece, what remains (the reaches
// with more than that upstream: the main stems) is the trunk.  Pieces go to the least-loaded part, largest first; the
// trunk and the pieces that share its part form the last part.  Every cut edge then runs from a piece into the
// trunk, so the part graph has depth two, and the cut reaches enter the last part near its outlets, where the lag
// pipeline gives them the most slack: a downstream GPU starts almost together with its upstream ones (the nested
// parts of the min-max cut below make chains four or five parts deep, each link costing its upstream part's whole
// pipeline skew).  Returns false when the trunk is too large for that (chain-like networks).
static bool partition_trunk(const std::vector<int32_t> &down, int32_t n_parts, int32_t *part_of, std::vector<int64_t> &sizes)
{
    const int64_t n = (int64_t)down.size();
    if (n_parts < 2 || n < 64 * (int64_t)n_parts) return false;
    std::vector<int64_t> sub(n, 1);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) sub[down[c]] += sub[c];      // downstream reaches have larger indices
    const int64_t s = std::max<int64_t>(1, n / (4 * (int64_t)n_parts));
    std::vector<int32_t> root_of(n, -1);
    std::vector<std::pair<int64_t, int32_t>> pieces;       // (size, root)
    int64_t trunk = 0;
    for (int64_t v = n - 1; v >= 0; --v) {
        if (sub[v] > s) { ++trunk; continue; }
        if (down[v] < 0 || sub[down[v]] > s) { root_of[v] = (int32_t)v; pieces.emplace_back(sub[v], (int32_t)v); }
        else root_of[v] = root_of[down[v]];
    }
    const int64_t even = (n + n_parts - 1) / n_parts;
    if (trunk > even / 2) return false;
    // the pieces that join the trunk farthest from the outlets would be needed first by the trunk's part (smallest lag):
    // they stay with the trunk, up to an even share; the rest is spread over the other parts, largest first
    std::vector<int32_t> dist(n);
    for (int64_t c = n - 1; c >= 0; --c) dist[c] = down[c] < 0 ? 0 : dist[down[c]] + 1;
    std::sort(pieces.begin(), pieces.end(), [&](const auto &a, const auto &b) {
        return dist[a.second] != dist[b.second] ? dist[a.second] > dist[b.second] : a.second < b.second; });
    sizes.assign(n_parts, 0);
    sizes[n_parts - 1] = trunk;
    std::vector<int32_t> part_of_root(n, -1);
    // tributaries too small to be worth a boundary series of their own stay with the trunk
    const int64_t tiny = std::max<int64_t>(2, s / 1024);     // at 8M reaches / 8 parts: 550 boundary series instead of 11,800
    for (const auto &pc : pieces)
        if (pc.first < tiny) { sizes[n_parts - 1] += pc.first; part_of_root[pc.second] = n_parts - 1; }
    std::vector<std::pair<int64_t, int32_t>> rest;
    for (const auto &pc : pieces) {      // deepest junctions first
        if (part_of_root[pc.second] >= 0) continue;
        if (sizes[n_parts - 1] + pc.first <= even) { sizes[n_parts - 1] += pc.first; part_of_root[pc.second] = n_parts - 1; }
        else rest.push_back(pc);
    }
    std::sort(rest.begin(), rest.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    for (const auto &pc : rest) {
        int32_t best = 0;
        for (int32_t b = 1; b < n_parts - 1; ++b) if (sizes[b] < sizes[best]) best = b;
        sizes[best] += pc.first;
        part_of_root[pc.second] = best;
    }
    if (*std::max_element(sizes.begin(), sizes.end()) > even + even / 8) return false;
    for (int64_t v = 0; v < n; ++v) part_of[v] = root_of[v] < 0 ? n_parts - 1 : part_of_root[root_of[v]];
    return true;
}

void partition_forest(const std::vector<int32_t> &down, int32_t n_parts, int32_t *part_of, std::vector<int64_t> &sizes)
{
    const int64_t n = (int64_t)down.size();
    sizes.clear();
    if (n == 0) return;
    if (partition_trunk(down, n_parts, part_of, sizes)) return;
    sizes.clear();
    std::vector<int32_t> up_ptr(n + 1, 0);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) ++up_ptr[down[c] + 1];
    for (int64_t i = 0; i < n; ++i) up_ptr[i + 1] += up_ptr[i];
    std::vector<int32_t> up_idx(up_ptr[n]), fill(up_ptr.begin(), up_ptr.end() - 1);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) up_idx[fill[down[c]]++] = (int32_t)c;

    Pieces P;
    std::vector<int32_t> bin_of, bin_depth;
    std::vector<int64_t> bin_size;
    int64_t lo = (n + n_parts - 1) / n_parts, hi = n;   // smallest feasible cap in [lo, hi]; hi is always feasible
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        cut_forest(down, up_ptr, up_idx, mid, P);
        if (pack_pieces(P, mid, bin_of, bin_size, bin_depth) <= n_parts) hi = mid; else lo = mid + 1;
    }
    cut_forest(down, up_ptr, up_idx, lo, P);
    const int64_t nb = pack_pieces(P, lo, bin_of, bin_size, bin_depth);
    // bins come out deepest class first = upstream-first, which is the numbering the callers want
    sizes.assign(bin_size.begin(), bin_size.end());
    (void)nb;
    for (int64_t v = 0; v < n; ++v) part_of[v] = bin_of[P.root_of[v]];
}


void build_tile_plan(const std::vector<int32_t> &down, const std::vector<int32_t> &lag_of, int32_t block, TilePlan &T,
                     const std::vector<uint8_t> *big_in, bool skeleton_only)
{
    T = TilePlan();
    T.block = block;
    const int64_t n = (int64_t)down.size();
    T.tile_ptr.assign(1, 0);
    T.level_start.assign(1, 0);
    if (n == 0) { T.ok = true; return; }
    std::vector<int32_t> up_ptr(n + 1, 0);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) ++up_ptr[down[c] + 1];
    for (int64_t i = 0; i < n; ++i) up_ptr[i + 1] += up_ptr[i];
    std::vector<int32_t> up_idx(up_ptr[n]), fill(up_ptr.begin(), up_ptr.end() - 1);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) up_idx[fill[down[c]]++] = (int32_t)c;
    auto is_hw = [&](int32_t i) { return up_ptr[i + 1] == up_ptr[i]; };

    std::vector<int64_t> sub(n, 1);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) sub[down[c]] += sub[c];
    auto big = [&](int64_t v) { return big_in ? (*big_in)[v] != 0 : sub[v] > block; };
    auto small_root = [&](int64_t v) { return !big(v) && (down[v] < 0 || big(down[v])); };

    std::vector<int32_t> tile_of_reach(n, -1);
    // ---- level 0: complete small subtrees, packed in order of their outlets' lag; a tile that the next subtree does not
    // fit is topped up from the following kWindow subtrees (largest fit first) before it is closed
    std::vector<int32_t> roots;
    if (!skeleton_only) for (int64_t v = 0; v < n; ++v) if (small_root(v)) roots.push_back((int32_t)v);
    std::stable_sort(roots.begin(), roots.end(), [&](int32_t a, int32_t b) { return lag_of[a] < lag_of[b]; });
    int32_t n_tiles = 0;
    {
        constexpr size_t kWindow = 768;
        std::vector<uint8_t> used(roots.size(), 0);
        size_t head = 0;
        while (head < roots.size()) {
            if (used[head]) { ++head; continue; }
            int64_t room = block;
            for (size_t i = head; i < roots.size() && room > 0; ++i) {      // in order while they fit
                if (used[i]) continue;
                if (sub[roots[i]] > room) break;
                used[i] = 1; tile_of_reach[roots[i]] = n_tiles; room -= sub[roots[i]];
            }
            while (room > 0) {                                              // top up: largest that still fits
                int64_t best = -1, best_size = 0;
                size_t seen = 0;
                for (size_t i = head; i < roots.size() && seen < kWindow; ++i) {
                    if (used[i]) continue;
                    ++seen;
                    if (sub[roots[i]] <= room && sub[roots[i]] > best_size) { best = (int64_t)i; best_size = sub[roots[i]]; }
                }
                if (best < 0) break;
                used[best] = 1; tile_of_reach[roots[best]] = n_tiles; room -= best_size;
            }
            ++n_tiles;
        }
    }
    const int32_t n_level0 = n_tiles;
    std::vector<int32_t> tile_level((size_t)n_tiles, 0);

    // ---- skeleton: weight = the reach + one ghost per small tributary; bottom-up cut into pieces of at most `block`
    std::vector<int32_t> bigs;
    for (int64_t v = 0; v < n; ++v) if (big(v)) bigs.push_back((int32_t)v);
    std::vector<int64_t> res(n, 0);
    std::vector<uint8_t> cut(n, 0);
    std::vector<std::pair<int64_t, int32_t>> kids;
    for (int32_t v : bigs) {
        int64_t total = 1;
        kids.clear();
        for (int32_t e = up_ptr[v]; e < up_ptr[v + 1]; ++e) {
            const int32_t c = up_idx[e];
            if (big(c)) { kids.emplace_back(res[c], c); total += res[c]; } else ++total;
        }
        if (total > block) {
            std::sort(kids.begin(), kids.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
            for (const auto &k : kids) {
                if (total <= block) break;
                cut[k.second] = 1;
                total -= k.first - 1;      // the piece leaves, its ghost stays
            }
        }
        if (total > block || up_ptr[v + 1] - up_ptr[v] > 65535) return;      // not tileable: T.ok stays false
        res[v] = total;
    }
    std::vector<int32_t> piece_root(n, -1), piece_level(n, 0);
    std::vector<int32_t> pieces;
    for (auto it = bigs.rbegin(); it != bigs.rend(); ++it) {
        const int32_t v = *it;
        if (down[v] < 0 || cut[v]) { piece_root[v] = v; pieces.push_back(v); piece_level[v] = 1; }
        else piece_root[v] = piece_root[down[v]];
    }
    for (int32_t v : bigs)      // upstream first: a cut reach is the outlet of its piece, whose level is final by now
        if (cut[v]) { const int32_t r = piece_root[down[v]]; piece_level[r] = std::max(piece_level[r], piece_level[v] + 1); }
    std::sort(pieces.begin(), pieces.end(), [&](int32_t a, int32_t b) {
        if (piece_level[a] != piece_level[b]) return piece_level[a] < piece_level[b];
        return lag_of[a] != lag_of[b] ? lag_of[a] < lag_of[b] : a < b; });
    {
        std::vector<int64_t> room;           // of the tiles of the current level
        int32_t cur = -1, first_tile = n_tiles;
        std::vector<int32_t> tile_of_piece(n, -1);
        for (int32_t r : pieces) {
            if (piece_level[r] != cur) { cur = piece_level[r]; first_tile = n_tiles; room.clear(); }
            size_t b = 0;
            for (; b < room.size(); ++b) if (room[b] >= res[r]) break;
            if (b == room.size()) { room.push_back(block); tile_level.push_back(cur); ++n_tiles; }
            room[b] -= res[r];
            tile_of_piece[r] = first_tile + (int32_t)b;
        }
        for (int32_t v : bigs) tile_of_reach[v] = tile_of_piece[piece_root[v]];
    }
    if (!skeleton_only)
        for (int64_t v = n - 1; v >= 0; --v)      // small subtrees inherit the tile of their outlet
            if (!big(v) && !small_root(v)) tile_of_reach[v] = tile_of_reach[down[v]];
    (void)n_level0;

    // ---- positions: breadth-first from each tile's outlets, upstream positions contiguous, headwater tributaries first
    T.n_tiles = n_tiles;
    T.tile_level = tile_level;
    T.n_levels = tile_level.empty() ? 0 : tile_level.back() + 1;
    T.level_start.assign((size_t)T.n_levels + 1, 0);
    for (int32_t t = 0; t < n_tiles; ++t) ++T.level_start[tile_level[t] + 1];
    for (int32_t l = 0; l < T.n_levels; ++l) T.level_start[l + 1] += T.level_start[l];
    std::vector<int32_t> root_ptr((size_t)n_tiles + 1, 0);
    auto tile_root = [&](int64_t v) { return down[v] < 0 || tile_of_reach[down[v]] != tile_of_reach[v]; };
    for (int64_t v = 0; v < n; ++v) if (tile_of_reach[v] >= 0 && tile_root(v)) ++root_ptr[tile_of_reach[v] + 1];
    for (int32_t t = 0; t < n_tiles; ++t) root_ptr[t + 1] += root_ptr[t];
    std::vector<int32_t> root_list(root_ptr[n_tiles]), rfill(root_ptr.begin(), root_ptr.end() - 1);
    for (int64_t v = n - 1; v >= 0; --v) if (tile_of_reach[v] >= 0 && tile_root(v)) root_list[rfill[tile_of_reach[v]]++] = (int32_t)v;    // outlet-most first

    T.inv.assign(n, -1);
    T.tile_ptr.assign((size_t)n_tiles + 1, 0);
    T.tile_lag_lo.assign(n_tiles, 0); T.tile_lag_hi.assign(n_tiles, 0);
    std::vector<int32_t> ghost_positions;
    for (int32_t t = 0; t < n_tiles; ++t) {
        const int64_t base = (int64_t)T.perm.size();
        T.tile_ptr[t] = (int32_t)base;
        for (int32_t k = root_ptr[t]; k < root_ptr[t + 1]; ++k) {
            T.perm.push_back(root_list[k]); T.lag.push_back(lag_of[root_list[k]]);
        }
        for (int64_t head = base; head < (int64_t)T.perm.size(); ++head) {
            const int32_t v = T.perm[head];
            if (T.lag[head] & kTileGhost) { T.cfirst.push_back((int32_t)head); T.ccnt.push_back(0u); continue; }
            T.inv[v] = (int32_t)head;
            uint32_t cnt = 0, hw = 0;
            T.cfirst.push_back((int32_t)T.perm.size());
            for (int pass = 0; pass < 2; ++pass)
                for (int32_t e = up_ptr[v]; e < up_ptr[v + 1]; ++e) {
                    const int32_t c = up_idx[e];
                    if (is_hw(c) != (pass == 0)) continue;
                    const bool ghost = tile_of_reach[c] != t;
                    T.perm.push_back(c);
                    T.lag.push_back(lag_of[c] | (ghost ? kTileGhost : 0));
                    if (ghost) ghost_positions.push_back((int32_t)T.perm.size() - 1);
                    ++cnt; if (pass == 0) ++hw;
                }
            T.ccnt.push_back(cnt | (hw << 16));
        }
        int32_t lo = std::numeric_limits<int32_t>::max(), hi = 0;
        for (int64_t p = base; p < (int64_t)T.perm.size(); ++p) { const int32_t l = T.lag[p] & (kTileExport - 1); lo = std::min(lo, l); hi = std::max(hi, l); }
        T.tile_lag_lo[t] = lo; T.tile_lag_hi[t] = hi;
        if ((int64_t)T.perm.size() - base > block) return;      // cannot happen by construction; refuse rather than overrun
    }
    T.np = (int64_t)T.perm.size();
    T.tile_ptr[n_tiles] = (int32_t)T.np;
    T.tile_flags.assign(n_tiles, 0);
    for (int32_t t = 0; t < n_tiles; ++t)
        for (int32_t p = T.tile_ptr[t]; p < T.tile_ptr[t + 1]; ++p)
            if ((T.ccnt[p] & 0xFFFFu) > 3) { T.tile_flags[t] = 1; break; }      // kTileWide (rr_kernels_tile.hpp)
    T.n_ghost = (int64_t)ghost_positions.size();
    T.xpos.assign(T.np, -1);
    T.tile_of.assign(T.np, 0);
    for (int32_t t = 0; t < n_tiles; ++t) for (int32_t p = T.tile_ptr[t]; p < T.tile_ptr[t + 1]; ++p) T.tile_of[p] = t;
    if (skeleton_only) T.ext_ghost.assign(n, -1);
    for (int32_t g : ghost_positions) {
        const int32_t src = T.inv[T.perm[g]];
        if (src < 0) { T.ext_ghost[T.perm[g]] = g; continue; }      // mirrors a reach outside the plan: whoever routes it writes the record
        T.xpos[g] = src; T.xpos[src] = g; T.lag[src] |= kTileExport;
    }
    T.ok = true;
}

void build_direct_plan(const std::vector<int32_t> &down, const std::vector<int32_t> &lag_of, int32_t lanes, int32_t wmax, int32_t skel_block,
                       DirectPlan &D, const std::vector<uint8_t> *ghost, const std::vector<int32_t> *export_slot)
{
    D = DirectPlan();
    D.lanes = lanes; D.wmax = wmax;
    const int64_t n = (int64_t)down.size();
    if (n == 0) { D.why = "empty network"; return; }
    if (lanes > 1023) { D.why = "more lanes than a lane number holds"; return; }
    // subtree size, height (levels) and smallest index; upstream reaches have smaller indices
    std::vector<int32_t> sub(n, 1), height(n, 1), first(n), indeg(n, 0);
    for (int64_t v = 0; v < n; ++v) first[v] = (int32_t)v;
    for (int64_t c = 0; c < n; ++c)
        if (down[c] >= 0) {
            const int32_t d = down[c];
            sub[d] += sub[c]; height[d] = std::max(height[d], height[c] + 1); first[d] = std::min(first[d], first[c]); ++indeg[d];
        }
    D.big.assign(n, 0);
    for (int64_t v = 0; v < n; ++v) D.big[v] = (sub[v] > lanes || height[v] > wmax) ? 1 : 0;
    auto small_root = [&](int64_t v) { return !D.big[v] && (down[v] < 0 || D.big[down[v]]); };
    auto is_ghost = [&](int64_t v) { return ghost && (*ghost)[v] != 0; };
    // A boundary ghost's column lies wherever the caller put it (first, in a partitioned network's parts), not next to the reach it flows
    // into, so that reach cannot be a lane: it and what lies downstream of it joins the skeleton (in the part that holds the main stems
    // that is where the ghosts enter anyway; the few reaches at the upper ends of the stems, whose own sub-basins were cut away, are added).
    if (ghost)
        for (int64_t v = 0; v < n; ++v)
            if (is_ghost(v)) for (int64_t w = down[v]; w >= 0 && !D.big[w]; w = down[w]) D.big[w] = 1;
    for (int64_t v = 0; v < n; ++v) {
        if (is_ghost(v) && (indeg[v] != 0 || down[v] < 0)) { D.why = "a boundary ghost is not a headwater with a downstream reach"; return; }
        if (export_slot && (*export_slot)[v] >= 0 && down[v] >= 0) { D.why = "a boundary export has a downstream reach in this plan"; return; }
        if (D.big[v]) continue;
        if (indeg[v] > 3) { D.why = "a reach of a small subtree has more than three upstream reaches"; return; }
        if (small_root(v) && first[v] != (int32_t)v - sub[v] + 1) { D.why = "a small subtree is not a contiguous range of the params order"; return; }
    }
    // tiles: consecutive units (a whole small subtree, or one hole) while the lanes and the lag window last
    D.delay.assign(n, 0); D.up3.assign(n, 0x3FFFFFFF); D.xinfo.assign(n, -1);
    std::vector<int32_t> tile_of(n, -1);
    int64_t c = 0;
    while (c < n) {
        const int64_t c0 = c;
        int32_t lo = std::numeric_limits<int32_t>::max(), hi = -1, senders = 0;
        while (c < n) {
            int64_t c1;
            int32_t ulo = lo, uhi = hi, us = senders;
            if (D.big[c]) { c1 = c + 1; ++us; }
            else {      // the small subtree that STARTS here: its outlet is the first small root at or after c whose range begins at c
                int64_t r = c;
                while (!small_root(r)) r = down[r];      // c is the first (deepest-first) reach of exactly one small subtree
                if (first[r] != c) { D.why = "internal: a small subtree does not start where the previous unit ended"; return; }
                c1 = r + 1;
                if (down[r] >= 0 && !is_ghost(r)) ++us;
                if (!is_ghost(r)) for (int64_t v = c; v < c1; ++v) { ulo = std::min(ulo, lag_of[v]); uhi = std::max(uhi, lag_of[v]); }      // (a ghost is a unit of one idle column)
            }
            if (c1 - c0 > lanes || (uhi >= 0 && uhi - ulo + 1 > wmax) || us > kDirectSenders) break;
            lo = ulo; hi = uhi; senders = us; c = c1;
        }
        if (c == c0) { D.why = "internal: a unit does not fit an empty tile"; return; }
        const int32_t t = D.n_tiles++;
        D.tile_c0.push_back((int32_t)c0); D.tile_nc.push_back((int32_t)(c - c0));
        D.tile_lag_lo.push_back(hi < 0 ? 0 : lo); D.tile_span.push_back(hi < 0 ? 0 : hi - lo);
        for (int64_t v = c0; v < c; ++v) {
            tile_of[v] = t;
            if (D.big[v]) { D.delay[v] = kDirectHole; ++D.n_holes; }
            else if (is_ghost(v)) D.delay[v] = kDirectHole;      // passed through, not routed: its value arrives in the skeleton's record ring
            else D.delay[v] = lag_of[v] - lo;
        }
    }
    // upstream lanes (a small reach's upstream reaches are in its subtree, hence in its tile): headwater tributaries first, then
    // ascending index -- the order in which k_tile and k_tick add them, so that the three kernels agree bit for bit
    std::vector<int32_t> filled(n, 0);
    for (int pass = 0; pass < 2; ++pass)
        for (int64_t u = 0; u < n; ++u) {
            const int32_t d = down[u];
            if (d < 0 || D.big[d] || (indeg[u] == 0) != (pass == 0)) continue;
            const int32_t lane = (int32_t)(u - D.tile_c0[tile_of[d]]);
            const int k = filled[d]++;
            D.up3[d] = (D.up3[d] & ~(0x3FF << (10 * k))) | (lane << (10 * k));
            if (pass == 0) D.up3[d] = (int32_t)((uint32_t)D.up3[d] + (1u << 30));      // bits 30, 31: how many of them are headwaters (UnitMuskingum keeps them apart)
        }
    // tall subtrees (long unbranched runs) make most of a chain-like network skeleton: the record path routes that better
    if (D.n_holes * 10 > n) { D.why = "more than a tenth of the reaches have subtrees too large or too tall for a direct tile"; return; }
    // the skeleton keeps records
    build_tile_plan(down, lag_of, skel_block, D.skel, &D.big, true);
    if (!D.skel.ok) { D.why = "the skeleton does not tile"; return; }
    for (int64_t v = 0; v < n; ++v) {
        if (D.big[v]) D.xinfo[v] = D.skel.inv[v];
        else if (small_root(v) && down[v] >= 0) {
            if (D.skel.ext_ghost[v] < 0) { D.why = "internal: an outlet below the skeleton has no ghost"; return; }
            if (is_ghost(v)) continue;      // the in-pass of the boundary series writes that ghost's record (rr_exec.hpp), no lane sends
            D.xinfo[v] = D.skel.ext_ghost[v]; ++D.n_exports;
        }
    }
    if (export_slot)
        for (int64_t v = 0; v < n; ++v)
            if ((*export_slot)[v] >= 0 && !D.big[v]) { D.delay[v] |= kDirectExport; D.xinfo[v] = (*export_slot)[v]; }      // (an outlet: it sends nothing, xinfo is free)
    D.send_ptr.assign((size_t)D.n_tiles + 1, 0);
    for (int32_t t = 0; t < D.n_tiles; ++t) {
        for (int32_t v = D.tile_c0[t]; v < D.tile_c0[t] + D.tile_nc[t]; ++v)
            if (D.xinfo[v] >= 0 && !(D.delay[v] & kDirectExport)) {      // the column's number among the tile's senders travels in its delay word
                D.delay[v] |= ((int32_t)(D.send_lane.size() - D.send_ptr[t]) + 1) << kDirectSenderShift;
                D.send_lane.push_back((v - D.tile_c0[t]) | (D.big[v] ? kDirectHole : 0));
            }
        D.send_ptr[t + 1] = (int32_t)D.send_lane.size();
        if (D.send_ptr[t + 1] - D.send_ptr[t] > kDirectSenders) { D.why = "internal: a tile has more senders than one wave forwards"; return; }
    }
    D.ok = true;
}

constexpr int64_t kPostorderSmall = 256;      // = the lanes of a direct tile (rr_exec.hpp: kDirectLanes)

bool postorder(const int64_t *down, int64_t n, int64_t *order)
{
    if (n == 0) return true;
    // upstream lists, then sub-basin sizes by peeling headwaters (no order is assumed)
    std::vector<int64_t> up_ptr(n + 1, 0), pending(n, 0), sub(n, 1);
    for (int64_t i = 0; i < n; ++i) {
        if (down[i] >= n || down[i] == i) return false;
        if (down[i] >= 0) ++up_ptr[down[i] + 1];
    }
    for (int64_t i = 0; i < n; ++i) { pending[i] = up_ptr[i + 1]; up_ptr[i + 1] += up_ptr[i]; }
    std::vector<int64_t> up_idx(up_ptr[n]), fill(up_ptr.begin(), up_ptr.end() - 1), queue;
    for (int64_t i = 0; i < n; ++i) if (down[i] >= 0) up_idx[fill[down[i]]++] = i;
    queue.reserve(n);
    for (int64_t i = 0; i < n; ++i) if (pending[i] == 0) queue.push_back(i);
    for (size_t h = 0; h < queue.size(); ++h) {
        const int64_t v = queue[h], d = down[v];
        if (d >= 0) { sub[d] += sub[v]; if (--pending[d] == 0) queue.push_back(d); }
    }
    if ((int64_t)queue.size() != n) return false;      // a cycle keeps its reaches pending
    // Small tributaries first (sub-basins of at most kPostorderSmall reaches -- what a tile of the direct row path holds --, the largest of
    // them first), then the large ones, the LARGEST LAST: the reach then follows its main tributary's own main stem directly, so the
    // reaches of a main stem are neighbours in the table and the small sub-basins that join it lie in one run before them.  On the direct row
    // path the stems' reaches are the holes that a separate pass patches into the output rows with 8-byte stores: side by side they share
    // 64-byte lines (1M-reach random network: 7,375 distinct lines a row against 26,192 with the largest tributary first, which put every
    // stem reach right behind its own small tributary: the holes' pass 44 us per 128 rows against 140, the year 157 ms against 186-195;
    // 5,264 column-range tiles against 4,986 -- profiles/r05_postorder_sibling_order.txt).
    for (int64_t v = 0; v < n; ++v)
        std::sort(up_idx.begin() + up_ptr[v], up_idx.begin() + up_ptr[v + 1], [&](int64_t a, int64_t b) {
            const bool la = sub[a] > kPostorderSmall, lb = sub[b] > kPostorderSmall;
            if (la != lb) return lb;                                          // small before large
            if (sub[a] != sub[b]) return la ? sub[a] < sub[b] : sub[a] > sub[b];      // large: ascending (the main stem last); small: descending
            return a < b; });
    int64_t k = 0;
    std::vector<std::pair<int64_t, int64_t>> stack;      // (reach, next tributary)
    for (int64_t root = 0; root < n; ++root) {
        if (down[root] >= 0) continue;
        stack.emplace_back(root, up_ptr[root]);
        while (!stack.empty()) {
            auto &top = stack.back();
            if (top.second < up_ptr[top.first + 1]) { const int64_t c = up_idx[top.second++]; stack.emplace_back(c, up_ptr[c]); }
            else { order[k++] = top.first; stack.pop_back(); }
        }
    }
    return k == n;
}

int build_host_plan(int64_t n, const int32_t *indptr, const int32_t *indices, HostPlan &P, std::string &err)
{
    if (n < 0 || (n > 0 && (!indptr))) { err = "rr_plan_create: null csc_indptr or negative n"; return RR_E_INVALID; }
    if (n > (int64_t)std::numeric_limits<int32_t>::max() - 1) { err = "rr_plan_create: n exceeds int32 range"; return RR_E_INVALID; }
    P = HostPlan();
    P.n = n;
    if (n == 0) { P.child_ptr.assign(1, 0); P.lag_start.assign(1, 0); P.identity = true; return RR_OK; }
    if (indptr[0] != 0) { err = "rr_plan_create: csc_indptr[0] must be 0"; return RR_E_INVALID; }

    // ---- validate + downstream pointer (tools.py:94-106 guarantees these for the reference) ----
    std::vector<int32_t> down(n, -1);
    P.edge_of.assign(n, -1);
    for (int64_t c = 0; c < n; ++c) {
        const int64_t a = indptr[c], b = indptr[c + 1];
        if (b < a) { err = "rr_plan_create: csc_indptr is not non-decreasing"; return RR_E_INVALID; }
        if (b - a > 1) {
            err = "rr_plan_create: reach " + std::to_string(c) + " has " + std::to_string(b - a) +
                  " downstream reaches; the engine routes forests (one downstream per reach) only";
            return RR_E_UNSUPPORTED;
        }
        if (b > a) {
            if (!indices) { err = "rr_plan_create: null csc_indices"; return RR_E_INVALID; }
            const int64_t r = indices[a];
            if (r >= n || r < 0) { err = "rr_plan_create: csc row index out of range"; return RR_E_INVALID; }
            if (r <= c) { err = "params_file must be topologically sorted upstream to downstream"; return RR_E_NOT_TOPOLOGICAL; }
            down[c] = (int32_t)r;
            P.edge_of[c] = (int32_t)a;
        }
    }
    P.n_edges = indptr[n];
    P.down = down;

    // ---- distance to outlet; downstream reaches have larger params indices, so walk backwards ----
    std::vector<int32_t> dist(n);
    int32_t dmax = 0;
    for (int64_t c = n - 1; c >= 0; --c) {
        dist[c] = down[c] < 0 ? 0 : dist[down[c]] + 1;
        dmax = std::max(dmax, dist[c]);
    }
    P.depth = dmax + 1;

    // ---- upstream lists (ascending params index), headwater flags ----
    std::vector<int32_t> up_ptr(n + 1, 0);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) ++up_ptr[down[c] + 1];
    for (int64_t i = 0; i < n; ++i) up_ptr[i + 1] += up_ptr[i];
    std::vector<int32_t> up_idx(P.n_edges), fill(up_ptr.begin(), up_ptr.end() - 1);
    for (int64_t c = 0; c < n; ++c) if (down[c] >= 0) up_idx[fill[down[c]]++] = (int32_t)c;
    auto is_hw = [&](int32_t i) { return up_ptr[i + 1] == up_ptr[i]; };

    // ---- breadth-first order from the outlets; siblings: headwaters first, then ascending index ----
    std::vector<int32_t> bfs;
    bfs.reserve(n);
    std::vector<int64_t> level_start;  // in bfs ranks, by distance D
    level_start.push_back(0);
    for (int64_t c = 0; c < n; ++c) if (down[c] < 0) bfs.push_back((int32_t)c);
    P.n_outlets = (int64_t)bfs.size();
    P.hw_children.assign(n, 0);
    std::vector<uint16_t> hwc_by_node(n, 0);
    size_t head = 0;
    while (head < bfs.size()) {
        const size_t level_end = bfs.size();
        level_start.push_back((int64_t)level_end);
        for (; head < level_end; ++head) {
            const int32_t v = bfs[head];
            int64_t nh = 0;
            for (int32_t e = up_ptr[v]; e < up_ptr[v + 1]; ++e) if (is_hw(up_idx[e])) { bfs.push_back(up_idx[e]); ++nh; }
            for (int32_t e = up_ptr[v]; e < up_ptr[v + 1]; ++e) if (!is_hw(up_idx[e])) bfs.push_back(up_idx[e]);
            if (nh > 65535) { err = "rr_plan_create: a reach has more than 65535 headwater tributaries"; return RR_E_UNSUPPORTED; }
            hwc_by_node[v] = (uint16_t)nh;
        }
    }
    // the loop pushes one sentinel past the last non-empty level
    while (level_start.size() > 1 && level_start[level_start.size() - 1] == level_start[level_start.size() - 2])
        level_start.pop_back();
    if ((int64_t)bfs.size() != n || (int64_t)level_start.size() != (int64_t)P.depth + 1) {
        err = "rr_plan_create: internal error, breadth-first order incomplete";
        return RR_E_INVALID;
    }

    // ---- engine positions: levels in DESCENDING distance, bfs order inside a level ----
    P.perm.resize(n); P.inv.resize(n); P.lag.resize(n);
    P.lag_start.assign(P.depth + 1, 0);
    int64_t base = 0;
    for (int32_t d = dmax; d >= 0; --d) {
        const int64_t a = level_start[d], b = level_start[d + 1];
        const int32_t lag = dmax - d;
        P.lag_start[lag] = base;
        P.widest_level = std::max(P.widest_level, b - a);
        for (int64_t r = a; r < b; ++r) {
            const int64_t p = base + (r - a);
            P.perm[p] = bfs[r];
            P.inv[bfs[r]] = (int32_t)p;
            P.lag[p] = lag;
        }
        base += b - a;
    }
    P.lag_start[P.depth] = n;

    P.child_ptr.resize(n + 1);
    P.child_ptr[0] = 0;
    P.identity = true;
    for (int64_t p = 0; p < n; ++p) {
        const int32_t v = P.perm[p];
        P.child_ptr[p + 1] = P.child_ptr[p] + (up_ptr[v + 1] - up_ptr[v]);
        P.hw_children[p] = hwc_by_node[v];
        if (is_hw(v)) ++P.n_headwaters;
        if (v != p) P.identity = false;
    }
    // self-check of the contiguity property the kernels rely on
    for (int64_t p = 0; p < n; ++p) {
        const int32_t v = P.perm[p];
        int32_t q = P.child_ptr[p];
        for (int pass = 0; pass < 2; ++pass)
            for (int32_t e = up_ptr[v]; e < up_ptr[v + 1]; ++e)
                if (is_hw(up_idx[e]) == (pass == 0)) {
                    if (P.inv[up_idx[e]] != q || P.lag[q] != P.lag[p] - 1) {
                        err = "rr_plan_create: internal error, upstream reaches not contiguous in engine order";
                        return RR_E_INVALID;
                    }
                    ++q;
                }
    }
    P.inner_pos.reserve(n - P.n_headwaters);
    for (int64_t i = 0; i < n; ++i) if (!is_hw((int32_t)i)) P.inner_pos.push_back(P.inv[i]);
    return RR_OK;
}

}  // namespace rr

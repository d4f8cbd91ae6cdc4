// engine_bp.cpp -- BP update (abstractbeliefpropagationcache.jl:223-259): default sweep order, level schedule, message launches.
#include "engine_internal.hpp"
#include "launch_util.hpp"

namespace tnqs {

// can the last absorption of a BP message be fused into the Gram?  (c64, d = 2, the first row leg r has chi_r = 32,
// kept leg <= 32, tiles of 64 fibers = (s, i_r) aligned)
static int fused_leg(const State* s, const SD& sd, int jo) {
    if (s->dtype != TNQS_C64 || !use_mfma() || sd.d != 2 || sd.z < 2) return -1;
    int r = (jo == 0) ? 1 : 0;
    if (sd.chi[r] != 32 || sd.chi[jo] > 32 || sd.chi[jo] < 8) return -1;
    return r;
}
// ---------------------------------------------------------------------------------------------------------------
// BP update  (abstractbeliefpropagationcache.jl:223-259; Gauss-Seidel over edge_sequence, executed level by level)
// ---------------------------------------------------------------------------------------------------------------
struct BPPlan {
    std::vector<int> seq;                       // directed edge ids in sequence order
    std::vector<std::vector<int>> levels;       // positions in seq grouped by dependency level
    std::vector<int> pos_of;                    // de -> position in seq or -1
    std::vector<int> level_of;                  // position -> level
    bool in_place = false;                      // duplicates in the sequence: strictly sequential, single buffer
};

// Default sweep order (the reference's default is NamedGraphs' forest-cover sequence, not available here; any sequence gives the
// same fixed point, abstractbeliefpropagationcache.jl:204-218).  The edges are decomposed into LINEAR FORESTS (disjoint simple paths);
// inside a forest the messages are ordered so that every message is computed from the OLD values of the other messages of the same
// forest: along a path v0..vk the hops v_i -> v_{i+1} are listed last hop first, the hops v_{i+1} -> v_i first hop first (a message
// u -> w depends on the message entering u through its other path edge, which therefore must come LATER in the sequence).  Level
// scheduling then puts a whole forest into one level: 2 levels per sweep on a square lattice (rows, columns), and both outgoing
// messages of a site inside a forest share one pair product.  It is an ordinary sequential Gauss-Seidel order.
// On periodic lattices the paths close: there the edge sets may contain cycles (path_cycle_sequence, default_sequence below) -- still a sequential order.
struct DSU { std::vector<int> p; explicit DSU(int n) : p(n) { std::iota(p.begin(), p.end(), 0); } int f(int x) { while (p[x] != x) x = p[x] = p[p[x]]; return x; }
             bool join(int a, int b) { a = f(a); b = f(b); if (a == b) return false; p[a] = b; return true; } };
// Forests: the reference's own default order (NamedGraphs forest_cover_edge_sequence: per component, post-order DFS edges towards the
// root, then their reverses in reverse order).  With it ONE sweep is exact on a tree -- which is what the reference's tree defaults
// (maxiter = 1, no tolerance; beliefpropagationcache.jl:39,110-113) rely on.  The linear-forest order below lists every message BEFORE
// the one it depends on (so that a forest is one level), i.e. information moves one hop per sweep: right for loopy graphs, where the
// fixed point is iterated anyway, wrong for the single sweep of a tree.
static std::vector<int> tree_sequence(const Graph& g) {
    std::vector<int> seq; std::vector<char> seen(g.nv, 0);
    for (int root = 0; root < g.nv; ++root) {
        if (seen[root] || g.nbr[root].empty()) continue;
        std::vector<std::pair<int, int>> post;                          // (child, parent)
        std::vector<std::pair<int, size_t>> stack{{root, 0}}; std::vector<int> par(1, -1);
        seen[root] = 1;
        while (!stack.empty()) {
            auto& top = stack.back(); const int x = top.first;
            bool pushed = false;
            while (top.second < g.nbr[x].size()) {
                const int y = g.nbr[x][top.second++];
                if (seen[y]) continue;
                seen[y] = 1; stack.push_back({y, 0}); par.push_back(x); pushed = true; break;
            }
            if (pushed) continue;
            if (par.back() >= 0) post.push_back({x, par.back()});
            stack.pop_back(); par.pop_back();
        }
        for (auto& e : post) seq.push_back(g.dedge(e.first, e.second));
        for (auto it = post.rbegin(); it != post.rend(); ++it) seq.push_back(g.dedge(it->second, it->first));
    }
    return seq;
}
// The reference's default order on ANY graph (beliefpropagationcache.jl:28: NamedGraphs forest_cover_edge_sequence, restated -- it is not under
// /root/reference; julia/replay_golden.jl checks the restatement against NamedGraphs' own): the edges are covered greedily by spanning forests
// (breadth-first from the first vertex that still has an uncovered edge, neighbours in ascending vertex id); per tree the edges towards the root in
// depth-first post-order, then their reverses in reverse order.  Selected with tnqs_bp_opts.n_sequence = -1; the same sequence as the host's
// graphs.py forest_cover_edge_sequence passed explicitly.
static std::vector<int> forest_cover_sequence(const Graph& g) {
    std::vector<char> remaining(g.ne, 1); int left = g.ne;
    std::vector<int> seq;
    while (left > 0) {
        std::vector<char> visited(g.nv, 0), used(g.ne, 0);
        for (int root = 0; root < g.nv; ++root) {
            if (visited[root]) continue;
            bool any = false; for (int e : g.nbr_e[root]) any = any || remaining[e];
            if (!any) continue;
            std::vector<std::vector<int>> children(g.nv);
            std::vector<int> queue{root}; visited[root] = 1;
            for (size_t qi = 0; qi < queue.size(); ++qi) {
                const int x = queue[qi];
                for (size_t j = 0; j < g.nbr[x].size(); ++j) {
                    const int y = g.nbr[x][j], e = g.nbr_e[x][j];
                    if (visited[y] || !remaining[e]) continue;
                    visited[y] = 1; children[x].push_back(y); used[e] = 1; queue.push_back(y);
                }
            }
            std::vector<std::pair<int, int>> post;                      // (child, parent)
            std::vector<std::pair<int, size_t>> stack{{root, 0}};
            while (!stack.empty()) {
                auto& top = stack.back();
                if (top.second < children[top.first].size()) { const int c = children[top.first][top.second++]; stack.push_back({c, 0}); }
                else { const int x = top.first; stack.pop_back(); if (!stack.empty()) post.push_back({x, stack.back().first}); }
            }
            for (auto& e : post) seq.push_back(g.dedge(e.first, e.second));
            for (auto it = post.rbegin(); it != post.rend(); ++it) seq.push_back(g.dedge(it->second, it->first));
        }
        for (int e = 0; e < g.ne; ++e) if (used[e]) { remaining[e] = 0; --left; }
    }
    return seq;
}
// dependency level of every position of a sequence: one more than the highest level among the EARLIER positions whose message enters the source
// (a later position is read in its old value); a level's messages are independent of each other
// `starts` (optional, ascending positions): the levels of the positions from a start on lie above everything before it (later is always allowed: which
// value a message reads is decided by the positions, not by the levels) -- a site then never meets messages of two sets in one level
static std::vector<int> sequence_levels(const Graph& g, const std::vector<int>& seq, const std::vector<int>& pos_of, const std::vector<int>* starts = nullptr) {
    std::vector<int> level(seq.size(), 0);
    int floor_lv = 0, top = -1; size_t ks = 0;
    for (size_t t = 0; t < seq.size(); ++t) {
        if (starts) while (ks < starts->size() && (*starts)[ks] == (int)t) { floor_lv = top + 1; ++ks; }
        int de = seq[t]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e]; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
        int lv = 0;
        for (size_t j = 0; j < g.nbr[src].size(); ++j) {
            int k = g.nbr[src][j]; if (k == dst) continue;
            int pp = pos_of[g.dedge(k, src)];
            if (pp >= 0 && pp < (int)t) lv = std::max(lv, level[pp] + 1);
        }
        lv = std::max(lv, floor_lv);
        level[t] = lv; top = std::max(top, lv);
    }
    return level;
}
// The messages of edge sets of maximum degree 2 (`part[e]` = set of edge e), set after set.  Along a path v0..vk the hops v_i -> v_{i+1} are listed
// last hop first and the hops v_{i+1} -> v_i first hop first: every message is computed from the OLD values of its own set, the whole path is one
// level and an inner site sends both its messages in it.  Round a cycle v0..v_{n-1} the same holds for all hops but the two that leave v0, which
// come last (one of the messages a cycle carries must see a new value in any sequential order): v1..v_{n-1} send both their messages in one level,
// v0 both of its own in the next -- n two-message passes per cycle, where a path plus its closing edge in another forest takes n + 2 passes.
static std::vector<int> path_cycle_sequence(const Graph& g, const std::vector<int>& part, int np, std::vector<int>& starts) {
    std::vector<int> seq; starts.clear();
    for (int f = 0; f < np; ++f) {
        starts.push_back((int)seq.size());
        std::vector<std::vector<int>> adj(g.nv);
        for (int e = 0; e < g.ne; ++e) if (part[e] == f) { adj[g.esrc[e]].push_back(g.edst[e]); adj[g.edst[e]].push_back(g.esrc[e]); }
        std::vector<char> seen(g.nv, 0);
        for (int v = 0; v < g.nv; ++v) {
            if (adj[v].size() != 1 || seen[v]) continue;                   // start at a path end
            std::vector<int> path{v}; seen[v] = 1; int prev = -1, cur = v;
            for (;;) { int nxt = -1; for (int w : adj[cur]) if (w != prev) nxt = w; if (nxt < 0) break; prev = cur; cur = nxt; path.push_back(cur); seen[cur] = 1; }
            const int k = (int)path.size() - 1;
            for (int i = k - 1; i >= 0; --i) seq.push_back(g.dedge(path[i], path[i + 1]));
            for (int i = 0; i < k; ++i) seq.push_back(g.dedge(path[i + 1], path[i]));
        }
        for (int v = 0; v < g.nv; ++v) {
            if (adj[v].size() != 2 || seen[v]) continue;                   // what is left has no end: cycles
            std::vector<int> cyc{v}; seen[v] = 1; int prev = -1, cur = v;
            for (;;) { int nxt = (adj[cur][0] != prev) ? adj[cur][0] : adj[cur][1]; if (nxt == v) break; prev = cur; cur = nxt; cyc.push_back(cur); seen[cur] = 1; }
            const int n = (int)cyc.size();
            for (int i = n - 1; i >= 1; --i) seq.push_back(g.dedge(cyc[i], cyc[(i + 1) % n]));
            for (int i = 0; i + 1 < n; ++i) seq.push_back(g.dedge(cyc[i + 1], cyc[i]));
            seq.push_back(g.dedge(cyc[0], cyc[1])); seq.push_back(g.dedge(cyc[0], cyc[n - 1]));
        }
    }
    return seq;
}
// edge sets of maximum degree 2; with_cycles = false: linear forests (no cycle closes inside a set)
static void degree2_sets(const Graph& g, bool with_cycles, std::vector<int>& part, int& np) {
    part.assign(g.ne, -1); np = 0;
    // 1. unions of two colour classes (straight lines on lattices): pair the colours up; a union is a set of paths and even cycles
    std::vector<std::vector<char>> ok(g.ncolors, std::vector<char>(g.ncolors, 0));
    for (int a = 0; a < g.ncolors; ++a) for (int b = a + 1; b < g.ncolors; ++b) {
        bool acyclic = true;
        if (!with_cycles) { DSU d(g.nv); for (int e = 0; e < g.ne && acyclic; ++e) if (g.ecolor[e] == a || g.ecolor[e] == b) acyclic = d.join(g.esrc[e], g.edst[e]); }
        ok[a][b] = ok[b][a] = acyclic ? 1 : 0;
    }
    std::vector<int> mate(g.ncolors, -1), best;
    int best_pairs = -1;
    std::function<void(int, int)> rec = [&](int c, int pairs) {          // maximum matching of the colours (few colours: brute force)
        while (c < g.ncolors && mate[c] >= 0) ++c;
        if (c >= g.ncolors) { if (pairs > best_pairs) { best_pairs = pairs; best = mate; } return; }
        mate[c] = c; rec(c + 1, pairs); mate[c] = -1;                     // leave c single
        for (int b = c + 1; b < g.ncolors; ++b) if (mate[b] < 0 && ok[c][b]) { mate[c] = b; mate[b] = c; rec(c + 1, pairs + 1); mate[c] = mate[b] = -1; }
    };
    if (g.ncolors <= 10) rec(0, 0);
    // 2. greedy: an edge joins the first set where both ends still have degree < 2 (and, for forests, no cycle closes)
    std::vector<int> gpart(g.ne, -1); int gnp = 0;
    {
        std::vector<std::vector<int>> deg; std::vector<DSU> comp;
        for (int e = 0; e < g.ne; ++e) {
            int a = g.esrc[e], b = g.edst[e], f = 0;
            for (;; ++f) {
                if (f == gnp) { deg.emplace_back(g.nv, 0); comp.emplace_back(g.nv); ++gnp; }
                if (deg[f][a] < 2 && deg[f][b] < 2 && (with_cycles || comp[f].f(a) != comp[f].f(b))) break;
            }
            comp[f].join(a, b); ++deg[f][a]; ++deg[f][b]; gpart[e] = f;
        }
    }
    // the decomposition with fewer sets wins; ties go to the colour pairs
    if (best_pairs > 0 && g.ncolors - best_pairs <= gnp) {
        std::vector<int> fof(g.ncolors, -1);
        for (int c = 0; c < g.ncolors; ++c) if (fof[c] < 0) { fof[c] = np; if (best[c] != c && best[c] >= 0) fof[best[c]] = np; ++np; }
        for (int e = 0; e < g.ne; ++e) part[e] = fof[g.ecolor[e]];
    } else { part = gpart; np = gnp; }
}
static std::vector<int> default_sequence(const Graph& g, std::vector<int>& set_starts) {
    set_starts.clear();
    if (g.is_tree) {
        std::vector<int> seq = tree_sequence(g);
        if ((int)seq.size() != 2 * g.ne) throw Err(TNQS_ERR_HIP, "internal: tree sequence does not cover every message");
        return seq;
    }
    // linear forests (a forest is one level), or -- where it saves at least a twentieth of the (site, level) passes over the site tensors, i.e. on periodic
    // lattices -- sets that may close cycles (two levels per set, see path_cycle_sequence).  A pass is what a sweep costs on big tensors; the levels are
    // what it costs on small ones, and there the forests have fewer
    std::vector<int> best_seq; long best_passes = -1;
    const int nvariants = 2;
    for (int with_cycles = 0; with_cycles < nvariants; ++with_cycles) {
        std::vector<int> part; int np = 0;
        degree2_sets(g, with_cycles != 0, part, np);
        std::vector<int> starts;
        std::vector<int> seq = path_cycle_sequence(g, part, np, starts);
        if ((int)seq.size() != 2 * g.ne) throw Err(TNQS_ERR_HIP, "internal: default sequence does not cover every message");
        std::vector<int> pos_of(2 * (size_t)g.ne, -1);
        for (size_t t = 0; t < seq.size(); ++t) pos_of[seq[t]] = (int)t;
        if (!with_cycles) starts.clear();                                 // forests: plain dependency levels, as ever
        const std::vector<int> level = sequence_levels(g, seq, pos_of, starts.empty() ? nullptr : &starts);
        std::vector<std::pair<int, int>> sl;
        for (size_t t = 0; t < seq.size(); ++t) { int de = seq[t]; int e = de / 2; sl.push_back({(de & 1) ? g.edst[e] : g.esrc[e], level[t]}); }
        std::sort(sl.begin(), sl.end()); sl.erase(std::unique(sl.begin(), sl.end()), sl.end());
        const long passes = (long)sl.size();
        if (best_passes < 0 || passes * 20 <= best_passes * 19) { best_seq = std::move(seq); best_passes = passes; set_starts = starts; }
    }
    return best_seq;
}

// the default order as (src, dst) vertex pairs, for tests that replay it on the oracle (include/tnqs_debug.h)
void dbg_default_sequence(const State* s, std::vector<int>& src, std::vector<int>& dst) {
    const Graph& g = *s->g;
    if (g.default_seq.empty() && g.ne > 0) g.default_seq = default_sequence(g, g.default_set_starts);
    for (int de : g.default_seq) { const int e = de / 2; src.push_back((de & 1) ? g.edst[e] : g.esrc[e]); dst.push_back((de & 1) ? g.esrc[e] : g.edst[e]); }
}

// the same from the graph alone, with the dependency levels bp_update schedules the order in (host only, no device: tests/test_bp_schedule.py)
void dbg_default_sequence_graph(const Graph& g, std::vector<int>& src, std::vector<int>& dst, std::vector<int>& level) {
    if (g.default_seq.empty() && g.ne > 0) g.default_seq = default_sequence(g, g.default_set_starts);
    std::vector<int> pos_of(2 * (size_t)g.ne, -1);
    for (size_t t = 0; t < g.default_seq.size(); ++t) pos_of[g.default_seq[t]] = (int)t;
    level = sequence_levels(g, g.default_seq, pos_of, g.default_set_starts.empty() ? nullptr : &g.default_set_starts);
    for (int de : g.default_seq) { const int e = de / 2; src.push_back((de & 1) ? g.edst[e] : g.esrc[e]); dst.push_back((de & 1) ? g.esrc[e] : g.edst[e]); }
}

static BPPlan make_plan(const State* s, const tnqs_bp_opts* o) {
    const Graph& g = *s->g;
    BPPlan p;
    if (o && o->n_sequence > 0) {
        for (int i = 0; i < o->n_sequence; ++i) {
            int de = g.dedge(o->seq_src[i], o->seq_dst[i]);
            if (de < 0) throw Err(TNQS_ERR_INVALID, "bp_update: edge_sequence contains a pair of non-adjacent vertices");
            p.seq.push_back(de);
        }
    } else if (o && o->n_sequence < 0) p.seq = forest_cover_sequence(g);      // the reference's own default order
    else { if (g.default_seq.empty() && g.ne > 0) g.default_seq = default_sequence(g, g.default_set_starts); p.seq = g.default_seq; }
    p.pos_of.assign(2 * (size_t)g.ne, -1);
    for (size_t t = 0; t < p.seq.size(); ++t) { if (p.pos_of[p.seq[t]] >= 0) p.in_place = true; p.pos_of[p.seq[t]] = (int)t; }
    if (p.in_place) { for (size_t t = 0; t < p.seq.size(); ++t) { p.levels.push_back({(int)t}); p.level_of.push_back((int)t); } return p; }
    const bool is_default = !(o && o->n_sequence != 0);
    std::vector<int> level = sequence_levels(g, p.seq, p.pos_of, is_default && !g.default_set_starts.empty() ? &g.default_set_starts : nullptr); int nlev = 0;
    for (int lv : level) nlev = std::max(nlev, lv + 1);
    p.levels.resize(nlev);
    for (size_t t = 0; t < p.seq.size(); ++t) p.levels[level[t]].push_back((int)t);
    // the messages of a level are independent of each other: list them by source vertex, so that a workspace-bounded sub-batch (bp_update_t)
    // holds all messages of the sites it touches (they share the pair product and the double pair-Gram pass)
    auto src_of = [&](int t) { int de = p.seq[t]; int e = de / 2; return (de & 1) ? g.edst[e] : g.esrc[e]; };
    for (auto& lev : p.levels) std::stable_sort(lev.begin(), lev.end(), [&](int a, int b) { return src_of(a) < src_of(b); });
    p.level_of = level;
    return p;
}

static double default_tol(const State* s) { return s->dtype == TNQS_C64 ? 1e-5 : 1e-8; }   // beliefpropagationcache.jl:104-108

// ---- shared pair products ---------------------------------------------------------------------------------------
// A degree-4 site sends four messages per sweep, each needing the other three incoming messages absorbed.  Its legs are
// split into two pairs {A, B} by the level at which their outgoing message is computed; for an outgoing leg in A the pair
// product T_B = psi x m_b1 x m_b2 is shared with the other leg of A (the messages entering through B do not change between
// the two levels of A in the level-scheduled sequences), so a sweep costs 2 pair products + 4 (absorb + Gram) passes instead of
// 4 + 4.  Validity is not assumed but checked: an entry is reused only while the very same site / message buffers are current.
struct SharedT { Buf site, ma, mb, T; int la = -1, lb = -1; };

// ---- partial products kept ACROSS levels (sites the plane kernels of the bulk shape do not cover) -------------------------------------
// A message leaving a site absorbs the messages on all its other legs; which of them are absorbed in two-leg passes and which one inside
// the Gram pass is free.  Two facts make most of those passes redundant on lattices whose forests are straight lines: (i) the levels of one
// axis (a path and the edge that closes it on a periodic lattice) need the SAME product over the other axes' legs -- only messages along
// the axis change in between; (ii) the products of two axes share the factor over the third axis' legs.  So every product a chain writes
// (the final one and the one before it: the ping-pong buffers are both intact) is remembered with the exact buffers it was built from,
// and a later chain continues from the largest remembered product whose (leg, message buffer) pairs are a subset of what it needs.
// Legs are absorbed most-stable first (the messages that stay unchanged for the most levels to come), so that the early products are the
// reusable ones and the leg whose message changes next is left for the Gram pass.  Validity is by buffer identity, never assumed: the
// entries hold references, so an address cannot be recycled while an entry names it.  3 x 3 x 3 periodic cubic lattice, chi = 16: 5
// two-leg passes per site and sweep instead of 6.8.
struct ProdEntry { Buf site; std::vector<std::pair<int, Buf>> legs; Buf prod; long long next_use = 0; };
// Which entries stay is decided by the level schedule, which is known in advance (round 5; least-recently-used before): `next_use` = the level, counted
// through the sweeps, at which the site next sends a message that can continue from the entry while every message it was built from is still current
// (bp_update_t: next_use_of).  A product without such a level is not stored at all, an entry that has served its last use is dropped when it is found,
// and what has to go -- more than `per_site` entries of a site, more bytes than `cap` -- is the entry whose next use is farthest away.
struct ProdCache {
    std::unordered_map<int, std::vector<ProdEntry>> by_site; int per_site = 3; size_t bytes = 0, cap = bp_cache_budget();
    int n_hits = 0, n_evicted = 0;          // diagnostics (tnqs_apply_stats): lookups that found a product; entries dropped by the per-site or the byte bound
    std::function<long long(int, const std::vector<std::pair<int, Buf>>&)> next_use_of;       // -1: never
    // the largest entry of site v whose legs all occur in `want` with the same buffer; returns false when there is none
    bool find(int v, const Buf& site, const std::vector<std::pair<int, Buf>>& want, ProdEntry& out) {
        auto it = by_site.find(v); if (it == by_site.end()) return false;
        ProdEntry* best = nullptr;
        for (auto& e : it->second) {
            if (e.site != site || (best && e.legs.size() <= best->legs.size())) continue;
            bool sub = true;
            for (auto& lm : e.legs) { bool f = false; for (auto& w : want) if (w.first == lm.first && w.second == lm.second) { f = true; break; } if (!f) { sub = false; break; } }
            if (sub) best = &e;
        }
        if (!best) return false;
        out = *best; ++n_hits;
        const long long nu = next_use_of ? next_use_of(v, best->legs) : 0;
        if (nu < 0) { bytes -= best->prod->bytes; it->second.erase(it->second.begin() + (best - it->second.data())); }      // its last use: the caller holds the buffer
        else best->next_use = nu;
        return true;
    }
    void put(int v, const Buf& site, std::vector<std::pair<int, Buf>> legs, const Buf& prod) {
        if (legs.size() < 2 || !prod) return;
        std::sort(legs.begin(), legs.end(), [](const std::pair<int, Buf>& a, const std::pair<int, Buf>& b) { return a.first < b.first; });
        const long long nu = next_use_of ? next_use_of(v, legs) : 0;
        if (nu < 0) return;
        auto& vec = by_site[v];
        for (auto& e : vec) if (e.site == site && e.legs == legs) { bytes += prod->bytes; bytes -= e.prod->bytes; e.prod = prod; e.next_use = nu; return; }
        if ((int)vec.size() >= per_site) {
            size_t far = 0; for (size_t i = 1; i < vec.size(); ++i) if (vec[i].next_use > vec[far].next_use) far = i;
            if (vec[far].next_use <= nu) return;                                       // everything kept is needed sooner than the new product
            bytes -= vec[far].prod->bytes; vec.erase(vec.begin() + (std::ptrdiff_t)far); ++n_evicted;
        }
        ProdEntry e; e.site = site; e.legs = std::move(legs); e.prod = prod; e.next_use = nu; bytes += prod->bytes; vec.push_back(std::move(e));
        if (bytes > cap) {                         // over the byte bound: the entries needed last go, in ONE pass -- down to 7/8 of the bound, so that a long
            // level under memory pressure does not rescan every entry for every product it stores (round-3 advisor finding)
            std::vector<std::pair<long long, int>> order;          // (next use, site)
            for (auto& kv : by_site) for (auto& en : kv.second) order.push_back({en.next_use, kv.first});
            std::sort(order.begin(), order.end(), [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first > b.first; });
            const size_t target = cap - cap / 8;
            for (auto& o : order) {
                if (bytes <= target) break;
                auto& vv = by_site[o.second];
                for (size_t i = 0; i < vv.size(); ++i) if (vv[i].next_use == o.first) { bytes -= vv[i].prod->bytes; vv.erase(vv.begin() + (std::ptrdiff_t)i); ++n_evicted; break; }
            }
        }
    }
};

template <class T> void bp_update_t(State* s, const tnqs_bp_opts* o, int* niter_out, double* diff_out, bool optimistic, int iters_before) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    // the level schedule depends on the graph and the sequence only: the one of the default sequence is kept with the graph
    std::shared_ptr<const BPPlan> plan_p;
    if (o && o->n_sequence > 0) plan_p = std::make_shared<const BPPlan>(make_plan(s, o));
    else if (o && o->n_sequence < 0) {       // the reference's default order: also a function of the graph alone
        if (!g.forest_plan) g.forest_plan = std::make_shared<const BPPlan>(make_plan(s, o));
        plan_p = std::static_pointer_cast<const BPPlan>(g.forest_plan);
    } else {
        if (!g.default_plan) g.default_plan = std::make_shared<const BPPlan>(make_plan(s, o));
        plan_p = std::static_pointer_cast<const BPPlan>(g.default_plan);
    }
    const BPPlan& plan = *plan_p;
    int maxiter = (o && o->maxiter > 0) ? o->maxiter : (g.is_tree ? 1 : 25);                  // :39,:103
    double tol;
    if (!o || std::isnan(o->tolerance)) tol = g.is_tree ? -1.0 : default_tol(s); else tol = o->tolerance;
    const bool compute_error = tol >= 0;
    const int normalize = o ? o->normalize : 1;
    if (!normalize) materialize_scale_all(s);      // un-normalised messages carry the absolute scale of the site tensors
    const size_t esz = s->esz();
    const size_t nseq = plan.seq.size();
    if (nseq == 0) { if (niter_out) *niter_out = 0; if (diff_out) *diff_out = 0; return; }
    PhaseScope phase_scope(s, TNQS_PROF_PHASE_BP_UPDATE); phase_scope.count = 0;       // launches = sweeps enqueued
    Buf d_diffs = dalloc(s, nseq * sizeof(double));
    Buf d_sum = dalloc(s, sizeof(double));
    std::vector<Buf> cur = s->msg;
    int niter = maxiter; double avg = 0; bool converged = false;
    // shared pair products (see SharedT): partner[v][j] = the leg paired with j, -1 when the site is not covered
    std::vector<std::array<int, 4>> partner(g.nv, std::array<int, 4>{{-1, -1, -1, -1}});
    std::vector<std::array<SharedT, 2>> tshare;
    if (std::is_same<T, float>::value && use_mfma() && use_pair()) {
        tshare.resize(g.nv);
        for (int v = 0; v < g.nv; ++v) {
            if (!s->owns(v) || g.nbr[v].size() != 4 || s->d[v] != 2) continue;
            bool ok = true; std::array<std::pair<int, int>, 4> ord;
            for (int j = 0; j < 4; ++j) {
                if (s->chi[g.nbr_e[v][j]] != 32) ok = false;
                int pp = plan.pos_of[g.dedge(v, g.nbr[v][j])];
                ord[j] = {pp >= 0 ? plan.level_of[pp] : INT_MAX, j};
            }
            if (!ok) continue;
            std::sort(ord.begin(), ord.end());
            partner[v][ord[0].second] = ord[1].second; partner[v][ord[1].second] = ord[0].second;
            partner[v][ord[2].second] = ord[3].second; partner[v][ord[3].second] = ord[2].second;
        }
    }
    ProdCache pcache;
    struct CacheStats { State* s; ProdCache& c; ~CacheStats() { s->stats.n_bp_products_reused += c.n_hits; s->stats.n_bp_products_evicted += c.n_evicted; } } cache_stats{s, pcache};
    {   // never more than a third of what the device has free right now (the products are an optimisation, the workspace is not)
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) pcache.cap = std::min(pcache.cap, (fr + s->pool->bytes_cached()) / 3);
    }
    const bool cache_on = use_prodcache() && !plan.in_place;
    const bool bra_on = use_bra_products() && s->msg_hermitian && !plan.in_place;
    const int nlev = (int)plan.levels.size();
    // levels until the message entering src through leg j changes again, seen from position t of the sequence (INT_MAX: never)
    auto horizon = [&](int src, int j, int t) -> int {
        const int pp = plan.pos_of[g.dedge(g.nbr[src][j], src)];
        if (pp < 0) return INT_MAX;
        const int Lc = plan.level_of[t], Lj = plan.level_of[pp];
        return Lj > Lc ? Lj - Lc : (Lj < Lc ? nlev - Lc + Lj : 0);
    };
    typedef std::vector<std::pair<int, Buf>> LegBufs;
    // the level (counted through the sweeps of this call) at which site v next sends a message that can continue from a product over `legs`: a message through
    // a leg outside `legs`, in a level where v sends nothing through a leg of `legs`, no later than the first level that recomputes a message entering through
    // `legs` (a message recomputed in the very level of the use is still read in its old value there).  -1: no such level
    int cur_level = 0, cur_iter = 1;
    std::vector<std::vector<int>> out_level(g.nv), in_level(g.nv);           // per site and leg: level of the outgoing / incoming message (-1: not in the sequence)
    for (int v = 0; v < g.nv; ++v) for (size_t j = 0; j < g.nbr[v].size(); ++j) {
        const int po = plan.pos_of[g.dedge(v, g.nbr[v][j])], pi = plan.pos_of[g.dedge(g.nbr[v][j], v)];
        out_level[v].push_back(po >= 0 ? plan.level_of[po] : -1); in_level[v].push_back(pi >= 0 ? plan.level_of[pi] : -1);
    }
    pcache.next_use_of = [&](int v, const LegBufs& legs) -> long long {
        if (plan.in_place) return 0;
        const int Lc = cur_level, z = (int)g.nbr[v].size();
        auto has = [&](int j) { for (auto& lm : legs) if (lm.first == j) return true; return false; };
        int life = INT_MAX;
        for (auto& lm : legs) { const int L = in_level[v][lm.first]; if (L < 0) continue; life = std::min(life, L > Lc ? L - Lc : (L < Lc ? nlev - Lc + L : 0)); }
        int best = INT_MAX;
        for (int j = 0; j < z; ++j) {
            const int L = out_level[v][j]; if (L < 0 || has(j)) continue;
            const int d = L > Lc ? L - Lc : nlev - Lc + L;
            if (d > life || d >= best) continue;
            bool clash = false; for (int j2 = 0; j2 < z; ++j2) if (out_level[v][j2] == L && has(j2)) clash = true;
            if (!clash) best = d;
        }
        if (best == INT_MAX) return -1;
        if (Lc + best >= nlev && cur_iter >= maxiter) return -1;              // the use lies in a sweep that will not happen
        return (long long)(cur_iter - 1) * nlev + Lc + best;
    };
    auto by_stability = [&](LegBufs& w, int src, int t) {
        std::stable_sort(w.begin(), w.end(), [&](const std::pair<int, Buf>& a, const std::pair<int, Buf>& b) { return horizon(src, a.first, t) > horizon(src, b.first, t); });
    };
    // every product a chain wrote that is still intact (its last two passes), with the buffers it was built from
    auto remember = [&](const Chain& c, const Buf& site, const LegBufs& base, const LegBufs& absorbed) {
        const int np = (int)c.trail.size();
        LegBufs acc = base;
        for (int k = 0; k < np; ++k) {
            for (int leg : c.trail[k]) for (auto& lm : absorbed) if (lm.first == leg) { acc.push_back(lm); break; }
            if (k >= np - 2 && c.tmp[k & 1]) pcache.put(c.v, site, acc, c.tmp[k & 1]);
        }
    };
    static const bool optimistic_on = !envflag("TNQS_NO_SPECULATION");
    // (sharded handles too, round 5: messages are replicated and every rank normalises / diffs all of them, so the verdict is the same on every rank)
    const bool go_optimistic = optimistic && optimistic_on && compute_error && iters_before == 0;
    for (int iter = 1 + iters_before; iter <= maxiter; ++iter) {
        phase_scope.count += 1;
        std::vector<Buf> fresh(2 * (size_t)g.ne);
        cur_iter = iter; cur_level = -1;
        for (auto& lev : plan.levels) {
            ++cur_level;
            // sub-batches bounded by workspace bytes
            size_t start = 0;
            while (start < lev.size()) {
                HostTimer ht_prep(0);
                size_t budget = bp_ws_budget(), used = 0, end = start;
                while (end < lev.size()) {
                    int de = plan.seq[lev[end]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                    size_t need = 2 * site_dims(s, src).n * esz;
                    if (end > start && used + need > budget) break;
                    used += need; ++end;
                }
                std::vector<Chain> chains; std::vector<int> tpos; std::vector<const void*> fmsg;
                std::unordered_map<size_t, LegBufs> cbase, cabs;           // chain index -> legs of the remembered product it starts from / legs it absorbs
                std::vector<PairItem> sh_pair; std::vector<PairGramItem> sh_gram; std::vector<int> sh_chain;   // shared-T path
                std::vector<int> is_shared_chain;
                std::vector<PairGram2Item> sh_dbl; std::vector<std::pair<int, int>> sh_dbl_chain;              // both messages of a forest in one pass
                // small sites (<= 8192 elements, ComplexF32): the whole message in ONE kernel with the tensor resident in LDS (kernels.hip bp_small_site_kernel);
                // TNQS_NO_SMALL_SITE_BP=1: the generic chain + Gram route
                std::vector<SmallMsgItem> small_items; std::vector<int> small_chain; int small_max = 0;
                static const bool small_on = !envflag("TNQS_NO_SMALL_SITE_BP");
                auto small_site = [&](int src) {
                    if (!small_on || !std::is_same<T, float>::value) return false;
                    const SD sd = site_dims(s, src);
                    return sd.n >= 64 && bp_small_site_covers(sd.d, sd.z, sd.chi.data(), sd.n);
                };
                struct Pend { int idx, jo, r; };                                                                // first message of a (site, T) seen in this level
                std::unordered_map<long long, Pend> pend;
                double sh_pair_slices = 0, sh_gram_slices = 0, sh_dbl_slices = 0;
                // ---- shared partial products for the sites the plane kernels do not cover (any degree, any bond dimension): a site that sends
                // several messages in this level absorbs the messages on its OTHER legs once (T = psi x_{legs not going out here} m) and every
                // outgoing message continues from T.  With the default linear-forest order a site sends two messages per level, so a degree-6
                // site does 4 + 2 x 1 absorption passes per level instead of 2 x 5.  Reuse is decided by buffer identity per message (the
                // Gauss-Seidel rule may give two messages of a site different versions of an incoming message), never assumed.
                struct Prefix { Buf site; std::vector<std::pair<int, const void*>> legs; Buf prod, bra; bool has_bra = false; };      // legs: of both products
                std::unordered_map<int, Prefix> prefix;
                std::vector<Buf> hits_alive;          // remembered products this sub-batch continues from: the cache may drop its entry (last use, or the byte bound) before the launches
                auto select_in = [&](int src, int j, int t) -> const Buf& {
                    int din = g.dedge(g.nbr[src][j], src); int pp = plan.pos_of[din];
                    return (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                };
                if (!plan.in_place) {
                    std::unordered_map<int, std::vector<int>> outl;            // source site -> legs going out in this sub-batch
                    auto generic_site = [&](int src, int jo) { return (tshare.empty() || site_dims(s, src).z != 4 || partner[src][jo] < 0) && !small_site(src); };
                    for (size_t q = start; q < end; ++q) {
                        int de = plan.seq[lev[q]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e]; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                        if (s->owns(src) && generic_site(src, g.leg(src, dst))) outl[src].push_back(g.leg(src, dst));
                    }
                    std::vector<Chain> pch; std::vector<int> psrc, pside; std::vector<LegBufs> pbase, pabs;
                    // one product of the level: the legs `w` of site src absorbed into psi (continuing from a remembered product when there is one); side 1 = the bra product
                    auto start_product = [&](int src, LegBufs w, int side) {
                        Prefix& pf = prefix[src];
                        Chain cp; cp.v = src; cp.src = s->site[src]->p; cp.sd = site_dims(s, src);
                        LegBufs base;
                        if (cache_on) {
                            cp.ordered = true;
                            ProdEntry hit;
                            if (pcache.find(src, s->site[src], w, hit)) {
                                base = hit.legs; cp.src = hit.prod->p; (side ? pf.bra : pf.prod) = hit.prod; hits_alive.push_back(hit.prod);      // (the product itself when nothing is left to absorb)
                                LegBufs rest; for (auto& x : w) { bool in = false; for (auto& b : base) in = in || b.first == x.first; if (!in) rest.push_back(x); }
                                w.swap(rest);
                            }
                        }
                        for (auto& x : w) cp.steps.push_back({x.first, x.second->p});
                        if (cp.steps.empty()) return;                                              // the whole product was remembered
                        pch.push_back(std::move(cp)); psrc.push_back(src); pside.push_back(side); pbase.push_back(std::move(base)); pabs.push_back(std::move(w));
                    };
                    for (size_t q = start; q < end; ++q) {
                        int t = lev[q]; int de = plan.seq[t]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        auto ol = outl.find(src);
                        if (ol == outl.end() || ol->second.size() < 2 || prefix.count(src)) continue;
                        const SD sd = site_dims(s, src);
                        Prefix pf; pf.site = s->site[src];
                        LegBufs want;
                        for (int j = 0; j < sd.z; ++j) {
                            if (std::find(ol->second.begin(), ol->second.end(), j) != ol->second.end()) continue;
                            const Buf& mb = select_in(src, j, t);
                            if (!mb) continue;
                            want.push_back({j, mb});
                        }
                        if (want.empty()) continue;
                        if (cache_on || bra_on) by_stability(want, src, t);
                        // ---- half of the messages on the BRA side (round 6).  m_out = sum (psi x_K m_k x_B m_b) conj(psi) with the messages of the legs B moved over:
                        // sum_b' m[b][b'] conj(psi[b']) = conj(sum_b' psi[b'] m[b'][b]) for a Hermitian m, i.e. conj(psi x_B m_b) -- the SAME two-leg product a ket
                        // side would use.  So the Gram pass takes X = psi x_K m_k and Y = psi x_B m_b, both one pass away from psi, instead of X = a product over
                        // K and B (two passes deep) and Y = psi; and the halves are split by how long their messages stay unchanged (an axis of a lattice each), so
                        // that the product over an axis is built once per sweep and serves first as the ket, then as the bra factor of the other axes' levels: a
                        // degree-6 site does 3 two-leg passes per sweep instead of 5.  Messages are Hermitian to rounding by construction (State::msg_hermitian).
                        const size_t nket = (bra_on && want.size() >= 4) ? (want.size() + 1) / 2 : want.size();
                        for (size_t i = 0; i < want.size(); ++i) pf.legs.push_back({want[i].first, want[i].second->p});
                        prefix[src] = pf;
                        start_product(src, LegBufs(want.begin(), want.begin() + (std::ptrdiff_t)nket), 0);
                        if (nket < want.size()) { prefix[src].has_bra = true; start_product(src, LegBufs(want.begin() + (std::ptrdiff_t)nket, want.end()), 1); }
                    }
                    if (!pch.empty()) {
                        run_chains<T>(s, pch, TNQS_PROF_BP_MODEPROD, TNQS_PROF_BP_PAIR);
                        for (size_t i = 0; i < pch.size(); ++i) {
                            Prefix& pf = prefix[psrc[i]];
                            for (int k = 0; k < 2; ++k) if (pch[i].tmp[k] && pch[i].tmp[k]->p == pch[i].result) (pside[i] ? pf.bra : pf.prod) = pch[i].tmp[k];
                            if (cache_on) remember(pch[i], pf.site, pbase[i], pabs[i]);
                        }
                    }
                }
                for (size_t q = start; q < end; ++q) {
                    int t = lev[q]; int de = plan.seq[t]; int e = de / 2;
                    int src = (de & 1) ? g.edst[e] : g.esrc[e]; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                    if (!s->owns(src)) continue;
                    Chain c; c.v = src; c.src = s->site[src]->p; c.sd = site_dims(s, src);
                    const int jo = g.leg(src, dst);
                    if (small_on && std::is_same<T, float>::value && c.sd.n >= 64 && bp_small_site_covers(c.sd.d, c.sd.z, c.sd.chi.data(), c.sd.n)) {
                        // small site: one kernel for the whole message (no shared products, no remembered ones: nothing of the generic bookkeeping below applies)
                        SmallMsgItem si{}; si.psi = c.src; si.d = c.sd.d; si.z = c.sd.z; si.jo = jo;
                        for (int j = 0; j < c.sd.z; ++j) {
                            si.chi[j] = c.sd.chi[j]; si.M[j] = nullptr;
                            const int k = g.nbr[src][j]; if (k == dst) continue;
                            const int din = g.dedge(k, src); const int pp = plan.pos_of[din];
                            const Buf& mb = (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                            if (mb) si.M[j] = mb->p;                 // unset message = identity: nothing to absorb
                        }
                        si.mfma = (c.sd.n % 256) == 0;
                        for (int j = 0; j < c.sd.z; ++j) if (c.sd.chi[j] != 16) si.mfma = 0;
                        small_items.push_back(si); small_chain.push_back((int)chains.size()); small_max = std::max(small_max, (int)c.sd.n);
                        is_shared_chain.push_back((int)chains.size());
                        chains.push_back(std::move(c)); tpos.push_back(t); fmsg.push_back(nullptr);
                        continue;
                    }
                    if (!tshare.empty() && c.sd.z == 4 && partner[src][jo] >= 0) {
                        const int r = partner[src][jo];
                        int pa = -1, pb = -1;
                        for (int j = 0; j < 4; ++j) if (j != jo && j != r) { if (pa < 0) pa = j; else pb = j; }
                        auto incoming = [&](int j) -> const Buf& {
                            int din = g.dedge(g.nbr[src][j], src); int pp = plan.pos_of[din];
                            return (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                        };
                        const Buf& ma = incoming(pa); const Buf& mb = incoming(pb); const Buf& mr = incoming(r);
                        PairGramItem gi{}; PairItem pi{};
                        if (ma && mb && mr && pair_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), pa, pb, pi.g)
                            && pair_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), r, jo, gi.g)) {
                            SharedT& sh = tshare[src][std::min(pa, pb) < std::min(r, jo) ? 0 : 1];      // slot of the pair {pa, pb}
                            if (!(sh.T && sh.site == s->site[src] && sh.ma == ma && sh.mb == mb && sh.la == pa && sh.lb == pb)) {
                                sh.site = s->site[src]; sh.ma = ma; sh.mb = mb; sh.la = pa; sh.lb = pb;
                                sh.T = dalloc(s, c.sd.n * esz);
                                pi.in = c.src; pi.out = sh.T->p; pi.Mx = ma->p; pi.My = mb->p;
                                sh_pair.push_back(pi); sh_pair_slices += (double)c.sd.n / 16384.0;
                            }
                            gi.X = sh.T->p; gi.Y = c.src; gi.M = mr->p;
                            const long long key = ((long long)src << 1) | (std::min(pa, pb) < std::min(r, jo) ? 0 : 1);
                            auto pit = pend.find(key);
                            if (pit != pend.end() && pit->second.jo == r && pit->second.r == jo && sh_gram[pit->second.idx].X == gi.X) {
                                // the partner message of the same forest is in this level too: one pass computes both
                                PairGramItem& first = sh_gram[pit->second.idx];       // plane (lx = r_first = jo, ly = jo_first = r)
                                PairGram2Item d2{}; d2.X = first.X; d2.Y = first.Y; d2.Mx = first.M; d2.My = gi.M; d2.g = first.g;
                                sh_dbl.push_back(d2); sh_dbl_chain.push_back({sh_chain[pit->second.idx], (int)chains.size()});
                                sh_dbl_slices += (double)c.sd.n / 8192.0;
                                first.X = nullptr;                                     // retired from the single list
                                sh_gram_slices -= (double)c.sd.n / 16384.0;
                                pend.erase(pit);
                            } else {
                                pend[key] = Pend{(int)sh_gram.size(), jo, r};
                                sh_gram.push_back(gi); sh_gram_slices += (double)c.sd.n / 16384.0;
                                sh_chain.push_back((int)chains.size());
                            }
                            is_shared_chain.push_back((int)chains.size());
                            chains.push_back(std::move(c)); tpos.push_back(t); fmsg.push_back(nullptr);
                            continue;
                        }
                    }
                    const int fr = fused_leg(s, c.sd, jo);
                    const void* fm = nullptr;
                    std::vector<char> done(c.sd.z, 0);                   // legs already absorbed in the shared partial product
                    {
                        auto pf = prefix.find(src);
                        if (pf != prefix.end() && pf->second.prod && (!pf->second.has_bra || pf->second.bra) && pf->second.site == s->site[src]) {
                            bool same = true;
                            for (auto& lm : pf->second.legs) { if (lm.first == jo) { same = false; break; } const Buf& mb = select_in(src, lm.first, t); if (!mb || mb->p != lm.second) { same = false; break; } }
                            if (same) { c.y = pf->second.has_bra ? pf->second.bra->p : c.src; c.src = pf->second.prod->p; for (auto& lm : pf->second.legs) done[lm.first] = 1; }
                        }
                    }
                    LegBufs want, base;
                    for (int j = 0; j < c.sd.z; ++j) {
                        int k = g.nbr[src][j]; if (k == dst || done[j]) continue;
                        int din = g.dedge(k, src); int pp = plan.pos_of[din];
                        const Buf& mb = (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                        if (!mb) continue;                               // unset message = identity: nothing to absorb
                        want.push_back({j, mb});
                    }
                    const bool from_prefix = c.y != nullptr;             // continues from this level's shared product: that product is remembered, not what follows
                    if (cache_on && !from_prefix) {
                        by_stability(want, src, t); c.ordered = true;
                        ProdEntry hit;
                        if (pcache.find(src, s->site[src], want, hit)) {
                            base = hit.legs; c.y = c.src; c.src = hit.prod->p; hits_alive.push_back(hit.prod);
                            LegBufs rest; for (auto& w : want) { bool in = false; for (auto& b : base) in = in || b.first == w.first; if (!in) rest.push_back(w); }
                            want.swap(rest);
                        }
                    }
                    for (auto& w : want) {
                        if (w.first == fr) fm = w.second->p;             // absorbed inside the Gram kernel
                        else c.steps.push_back({w.first, w.second->p});
                    }
                    if (cache_on && !from_prefix) { cbase[chains.size()] = std::move(base); cabs[chains.size()] = std::move(want); }
                    chains.push_back(std::move(c)); tpos.push_back(t); fmsg.push_back(fm);
                }
                // ---- 16-dimensional planes: the two messages a site sends in this level, both continuing from the same shared product and
                // each absorbing exactly the other's outgoing leg, come from ONE pass over (T, psi) (mfma_pair_gram2x16_kernel) ------------
                std::vector<PairGram2x16Item> g16; std::vector<std::pair<int, int>> g16_chain;       // (chain of the message through ly, through lx)
                if (std::is_same<T, float>::value && use_mfma() && use_pair()) {
                    std::unordered_map<int, std::vector<int>> by_src;
                    for (size_t ci = 0; ci < chains.size(); ++ci)
                        if (chains[ci].y && chains[ci].steps.size() == 1 && !fmsg[ci] && chains[ci].sd.n >= (size_t)(1u << 14)) by_src[chains[ci].v].push_back((int)ci);
                    for (auto& kv : by_src) {
                        if (kv.second.size() != 2) continue;
                        const int ci = kv.second[0], cj = kv.second[1];
                        Chain& a = chains[ci]; Chain& b = chains[cj];
                        auto out_leg = [&](int c) { int de = plan.seq[tpos[c]]; int e = de / 2; int dst = (de & 1) ? g.esrc[e] : g.edst[e]; return g.leg(chains[c].v, dst); };
                        const int ly = out_leg(ci), lx = out_leg(cj);
                        if (a.src != b.src || a.y != b.y || a.steps[0].first != lx || b.steps[0].first != ly) continue;
                        PairGram2x16Item it{};
                        if (!plane_geometry(a.sd.d, a.sd.z, a.sd.chi.data(), lx, ly, 16, it.g)) continue;
                        it.X = a.src; it.Y = a.y; it.Mx = a.steps[0].second; it.My = b.steps[0].second;
                        g16.push_back(it); g16_chain.push_back({ci, cj});
                        a.steps.clear(); b.steps.clear();
                        is_shared_chain.push_back(ci); is_shared_chain.push_back(cj);
                    }
                }
                // a site that sends ONE message in this level (no shared product): its last absorption is fused with the Gram too -- the chain
                // stops one leg early and the same kernel computes (T x_lx M) conj(psi) for that single message (My = null)
                std::vector<int> g16_single;                          // index into g16 of the single items (their chain runs first, X is set after it)
                if (std::is_same<T, float>::value && use_mfma() && use_pair()) {
                    for (size_t ci = 0; ci < chains.size(); ++ci) {
                        Chain& c = chains[ci];
                        if ((c.y && (!cbase.count(ci) || cbase[ci].empty())) || fmsg[ci] || c.steps.empty() || (c.steps.size() & 1) == 0 || c.sd.n < (size_t)(1u << 14)) continue;
                        bool sh = false; for (int q : is_shared_chain) sh = sh || q == (int)ci;
                        if (sh) continue;
                        const int de = plan.seq[tpos[ci]]; const int e = de / 2; const int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                        const int ly = g.leg(c.v, dst), lx = c.steps.back().first;
                        PairGram2x16Item it{};
                        if (!plane_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), lx, ly, 16, it.g)) continue;
                        bool all16 = true; for (auto& st : c.steps) all16 = all16 && c.sd.chi[st.first] == 16;
                        if (!all16) continue;
                        it.Y = c.y ? c.y : c.src; it.Mx = c.steps.back().second; it.My = nullptr; it.X = nullptr;      // X = the chain's result, known after run_chains
                        c.steps.pop_back();
                        g16_single.push_back((int)g16.size());
                        g16.push_back(it); g16_chain.push_back({(int)ci, -1});
                    }
                }
                ht_prep.stop();
                std::vector<char> is_shared(chains.size(), 0);
                for (int ci : is_shared_chain) is_shared[ci] = 1;
                HostTimer ht_launch(1);
                // Two independent launch chains make up a level: the bulk sites' plane kernels (pair product -> both-messages pair-Gram) and the other
                // sites' single-leg products -> Grams (boundary sites of a lattice: 24 of the 49 sites of a 7 x 7 one, small launches of 10-50 us each).
                // They meet in msg_finalize.  The second chain goes to the side stream, under the plane kernels (the chi = 16
                // pair-Gram items that continue from a chain's result keep everything on one stream); nothing released inside the region is handed out
                // again before the join (Pool::set_defer).
                hipStream_t const main_stream = s->stream;
                bool has_other = !small_items.empty(); for (size_t ci = 0; ci < chains.size(); ++ci) has_other = has_other || !is_shared[ci];
                const bool split_level = (!sh_pair.empty() || !sh_dbl.empty()) && has_other && g16.empty();
                hipStream_t side_stream = nullptr;
                if (split_level) {
                    side_stream = aux_stream_of(s);
                    HIPCHK(hipEventRecord(s->ev_fork, main_stream)); HIPCHK(hipStreamWaitEvent(side_stream, s->ev_fork, 0));
                    s->pool->set_defer(true);
                }
                struct SplitGuard { State* s; hipStream_t m; bool on; ~SplitGuard() { s->stream = m; if (on) s->pool->set_defer(false); } } split_guard{s, main_stream, split_level};
                auto on_side = [&](bool side) { if (split_level) { s->stream = side ? side_stream : main_stream; s->prof->chain = false; } };
                if (!sh_pair.empty()) {
                    const int spw = pair_spw(sh_pair_slices); int wgs = 0;
                    for (auto& it : sh_pair) { it.spw = spw; it.slice_begin = wgs; wgs += pair_wgs(it.g.n0 * it.g.n1 * it.g.n2, spw); }
                    const PairItem* d = upload(s, sh_pair);
                    ProfScope ps(s, TNQS_PROF_BP_PAIR, 2.0 * sh_pair_slices * 16384.0 * esz, 2 * 8.0 * sh_pair_slices * 16384.0 * 32);
                    launch_mfma_pair(s->stream, d, (int)sh_pair.size(), wgs);
                }
                on_side(true);
                run_chains<T>(s, chains, TNQS_PROF_BP_MODEPROD, TNQS_PROF_BP_PAIR);
                on_side(false);
                if (cache_on) for (auto& kv : cabs) if (!chains[kv.first].trail.empty()) remember(chains[kv.first], s->site[chains[kv.first].v], cbase[kv.first], kv.second);
                std::vector<GramJob> jobs;
                for (size_t i = 0; i < chains.size(); ++i) {
                    int de = plan.seq[tpos[i]]; int e = de / 2; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                    GramJob j{}; j.X = chains[i].result; j.Y = chains[i].y ? chains[i].y : chains[i].src; j.sd = chains[i].sd; j.leg = g.leg(chains[i].v, dst); j.keep_site = false;
                    j.M = fmsg[i];
                    jobs.push_back(j);
                }
                if (!small_items.empty()) {
                    size_t slab_bytes = 0;                   // one allocation for the raw messages of the level (views into it: a pool round trip per message otherwise)
                    for (size_t q = 0; q < small_items.size(); ++q) { const int co = small_items[q].chi[small_items[q].jo]; slab_bytes += round256((size_t)co * co * esz); }
                    Buf slab = dalloc(s, slab_bytes); size_t off = 0;
                    for (size_t q = 0; q < small_items.size(); ++q) {
                        GramJob& j = jobs[small_chain[q]];
                        const int co = small_items[q].chi[small_items[q].jo];
                        j.nchunks = 1; j.KK = co; j.partial = sub_buffer(slab, off, (size_t)co * co * esz); off += round256((size_t)co * co * esz);
                        small_items[q].out = j.partial->p;
                    }
                    // matrix-core form on an unsharded handle: the kernel holds the whole message and finishes it (normalisation, message_diff) -- one launch less on
                    // the critical path of the level
                    if (!s->sharded() && !plan.in_place) {
                        size_t nb_bytes = 0;
                        for (size_t q = 0; q < small_items.size(); ++q) if (small_items[q].mfma) nb_bytes += round256((size_t)256 * esz);
                        if (nb_bytes) {
                            Buf nslab = dalloc(s, nb_bytes); size_t noff = 0;
                            for (size_t q = 0; q < small_items.size(); ++q) {
                                SmallMsgItem& si = small_items[q]; if (!si.mfma) continue;
                                GramJob& j = jobs[small_chain[q]];
                                const int t = tpos[small_chain[q]]; const int de = plan.seq[t];
                                j.final_msg = sub_buffer(nslab, noff, (size_t)256 * esz); noff += round256((size_t)256 * esz);
                                si.new_msg = j.final_msg->p; si.old_msg = cur[de] ? cur[de]->p : nullptr;
                                si.diff_out = reinterpret_cast<double*>(d_diffs->p) + t; si.normalize = normalize;
                            }
                        }
                    }
                    on_side(true);          // (with a split level: next to the bulk sites' plane kernels, like the other boundary-site work; the descriptor copy
                                            //  travels on the same stream as the kernel that reads it)
                    const SmallMsgItem* d = upload_small(s, small_items);          // (one workgroup per item reads its own descriptor: straight from the pinned arena)
                    { ProfScope ps(s, TNQS_PROF_BP_FUSED, 0, 0); launch_bp_small_site(s->stream, d, (int)small_items.size(), small_max); }
                    on_side(false);
                }
                if (!sh_dbl.empty()) {
                    // full slices per workgroup pair: the largest power of two that still gives >= 4 workgroups per CU; an item gets groups
                    // of 16 workgroups (8 pairs), so powers of two avoid idle workgroups for the usual 2^k slices per site
                    int spw = 16, wgs = 0;
                    for (; spw > 1; spw >>= 1) {
                        long tot = 0;
                        for (auto& it : sh_dbl) { int np = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw; tot += 16 * ((np + 7) / 8); }      // (counted in half-slice workgroups for both kernels)
                        if (tot >= 1024) break;
                    }
                    for (size_t q = 0; q < sh_dbl.size(); ++q) {
                        PairGram2Item& it = sh_dbl[q]; GramJob& jy = jobs[sh_dbl_chain[q].first]; GramJob& jx = jobs[sh_dbl_chain[q].second];
                        const int npairs = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw;     // workgroup pairs (one per half), in groups of 8 pairs
                        const int nwg = pair_gram2_group() * ((npairs + 7) / 8);
                        it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        jy.nchunks = jx.nchunks = nwg; jy.KK = jx.KK = 32;             // one partial per workgroup
                        jy.partial = dalloc(s, (size_t)jy.nchunks * 1024 * esz); jx.partial = dalloc(s, (size_t)jx.nchunks * 1024 * esz);
                        it.partial_y = jy.partial->p; it.partial_x = jx.partial->p;
                    }
                    const PairGram2Item* d = upload(s, sh_dbl);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, 2.0 * sh_dbl_slices * 8192.0 * esz, 4 * 8.0 * sh_dbl_slices * 8192.0 * 32);
                    launch_mfma_pair_gram2(s->stream, d, (int)sh_dbl.size(), wgs);
                }
                for (int q : g16_single) { g16[q].X = chains[g16_chain[q].first].result; is_shared[g16_chain[q].first] = 1; }
                if (!g16.empty()) {
                    double tot = 0; for (auto& it : g16) tot += it.g.nslices();
                    int spw = pair_gram2x16_slices_at_a_time(); while (spw < 128 && tot / (2 * spw) >= 2048.0) spw *= 2;       // a multiple of the slices a workgroup walks at a time (waves / 2: half slices)
                    int wgs = 0; double by = 0, fl = 0;
                    for (size_t q = 0; q < g16.size(); ++q) {
                        PairGram2x16Item& it = g16[q]; GramJob& jy = jobs[g16_chain[q].first];
                        const int nwg = (it.g.nslices() + spw - 1) / spw;
                        it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        jy.nchunks = nwg; jy.KK = 16; jy.partial = dalloc(s, (size_t)nwg * 256 * esz); it.partial_y = jy.partial->p;
                        if (g16_chain[q].second >= 0) {
                            GramJob& jx = jobs[g16_chain[q].second];
                            jx.nchunks = nwg; jx.KK = 16; jx.partial = dalloc(s, (size_t)nwg * 256 * esz); it.partial_x = jx.partial->p;
                        } else it.partial_x = nullptr;
                        by += 2.0 * jy.sd.n * esz; fl += (g16_chain[q].second >= 0 ? 4 : 2) * 8.0 * jy.sd.n * 16;
                    }
                    const PairGram2x16Item* d = upload(s, g16);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, by, fl);
                    launch_mfma_pair_gram2x16(s->stream, d, (int)g16.size(), wgs);
                }
                {   // singles: drop the entries that were merged into a double item
                    std::vector<PairGramItem> keep; std::vector<int> keepc;
                    for (size_t q = 0; q < sh_gram.size(); ++q) if (sh_gram[q].X) { keep.push_back(sh_gram[q]); keepc.push_back(sh_chain[q]); }
                    sh_gram.swap(keep); sh_chain.swap(keepc);
                }
                if (!sh_gram.empty() && mfma_use_x3()) {
                    // the bf16 kernel in its one-message form (round 5): half-slice workgroups in groups of 16, one partial per workgroup.  These launches are what a
                    // sweep in the reference's forest-cover order consists of (a handful of messages per dependency level)
                    double tot = 0; for (auto& it : sh_gram) tot += (double)(it.g.n0 * it.g.n1 * it.g.n2);
                    int spw = 16; while (spw > 1 && 2.0 * tot / spw < 1024.0) spw >>= 1;
                    std::vector<PairGram2Item> one(sh_gram.size()); int wgs = 0;
                    for (size_t q = 0; q < sh_gram.size(); ++q) {
                        const PairGramItem& a = sh_gram[q]; PairGram2Item& it = one[q]; GramJob& j = jobs[sh_chain[q]];
                        const int np = (a.g.n0 * a.g.n1 * a.g.n2 + spw - 1) / spw, nwg = 16 * ((np + 7) / 8);
                        it.X = a.X; it.Y = a.Y; it.Mx = a.M; it.My = nullptr; it.g = a.g; it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        j.nchunks = nwg; j.KK = 32; j.partial = dalloc(s, (size_t)j.nchunks * 1024 * esz);
                        it.partial_y = j.partial->p; it.partial_x = nullptr;
                    }
                    const PairGram2Item* d = upload(s, one);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, 2.0 * sh_gram_slices * 16384.0 * esz, 2 * 8.0 * sh_gram_slices * 16384.0 * 32);
                    launch_x3_pair_gram1(s->stream, d, (int)one.size(), wgs);
                } else if (!sh_gram.empty()) {
                    int spw = (int)std::max(4.0, std::min(16.0, sh_gram_slices / 2048.0)); int wgs = 0;
                    for (size_t q = 0; q < sh_gram.size(); ++q) {
                        PairGramItem& it = sh_gram[q]; GramJob& j = jobs[sh_chain[q]];
                        int nwg = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw;
                        it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        j.nchunks = nwg; j.KK = 32; j.partial = dalloc(s, (size_t)j.nchunks * 1024 * esz);
                        it.partial = j.partial->p;
                    }
                    const PairGramItem* d = upload(s, sh_gram);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, 2.0 * sh_gram_slices * 16384.0 * esz, 2 * 8.0 * sh_gram_slices * 16384.0 * 32);
                    launch_mfma_pair_gram(s->stream, d, (int)sh_gram.size(), wgs);
                }
                {   // the fused and the plain Gram are different kernels: run them as two batches, keep the job order
                    std::vector<GramJob> jf, jp; std::vector<size_t> idf, idp;
                    for (size_t i = 0; i < jobs.size(); ++i) { if (is_shared[i]) continue; if (jobs[i].M) { jf.push_back(jobs[i]); idf.push_back(i); } else { jp.push_back(jobs[i]); idp.push_back(i); } }
                    on_side(true);
                    run_grams<T, T>(s, jf, TNQS_PROF_BP_FUSED);
                    run_grams<T, T>(s, jp, TNQS_PROF_BP_GRAM);
                    if (split_level) { HIPCHK(hipEventRecord(s->ev_join, side_stream)); HIPCHK(hipStreamWaitEvent(main_stream, s->ev_join, 0)); }
                    on_side(false);
                    for (size_t q = 0; q < jf.size(); ++q) jobs[idf[q]] = jf[q];
                    for (size_t q = 0; q < jp.size(); ++q) jobs[idp[q]] = jp[q];
                }
                ht_launch.stop();
                HostTimer ht_fin(2);
                std::vector<MsgFinalItem> fin;
                if (!s->sharded()) {
                    for (size_t i = 0; i < jobs.size(); ++i) {
                        int t = tpos[i]; int de = plan.seq[t]; int c = s->chi[de / 2];
                        if (jobs[i].final_msg) { fresh[de] = jobs[i].final_msg; continue; }
                        Buf nb = dalloc(s, (size_t)c * c * esz);
                        MsgFinalItem f{}; f.partial = jobs[i].partial->p; f.nchunks = jobs[i].nchunks; f.chi = c;
                        const Buf& oldb = plan.in_place && fresh[de] ? fresh[de] : cur[de];
                        f.old_msg = oldb ? oldb->p : nullptr; f.new_msg = nb->p;
                        f.diff_out = reinterpret_cast<double*>(d_diffs->p) + t; f.normalize = normalize;
                        fin.push_back(f);
                        fresh[de] = nb;
                    }
                } else {
                    // sharded: owners reduce their raw messages into the exchange buffer, all-gather, then EVERY rank
                    // normalises / diffs every message of the sub-batch (messages are replicated)
                    std::vector<size_t> slot(end - start, 0); std::vector<size_t> rank_bytes(s->nranks, 0);
                    for (size_t q = start; q < end; ++q) {
                        int de = plan.seq[lev[q]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        int r = s->owner[src]; slot[q - start] = rank_bytes[r];
                        rank_bytes[r] += round256((size_t)s->chi[e] * s->chi[e] * esz);
                    }
                    size_t stride = 0; for (size_t b : rank_bytes) stride = std::max(stride, b);
                    check_exchange(s, stride);
                    char* base = reinterpret_cast<char*>(s->exch);
                    std::vector<ReduceItem> ri; int elems = 0; size_t oi = 0;
                    for (size_t q = start; q < end; ++q) {
                        int de = plan.seq[lev[q]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        if (!s->owns(src)) continue;
                        int n2 = s->chi[e] * s->chi[e];
                        ri.push_back(ReduceItem{jobs[oi].partial->p, base + (size_t)s->rank * stride + slot[q - start], n2, jobs[oi].nchunks, 0, elems});
                        elems += n2; ++oi;
                    }
                    if (!ri.empty()) { const ReduceItem* dr = upload(s, ri); launch_reduce<T, T>(s->stream, dr, (int)ri.size(), elems); }
                    exchange(s, stride);
                    for (size_t q = start; q < end; ++q) {
                        int t = lev[q]; int de = plan.seq[t]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        int c = s->chi[e];
                        Buf nb = dalloc(s, (size_t)c * c * esz);
                        MsgFinalItem f{}; f.partial = base + (size_t)s->owner[src] * stride + slot[q - start]; f.nchunks = 1; f.chi = c;
                        const Buf& oldb = plan.in_place && fresh[de] ? fresh[de] : cur[de];
                        f.old_msg = oldb ? oldb->p : nullptr; f.new_msg = nb->p;
                        f.diff_out = reinterpret_cast<double*>(d_diffs->p) + t; f.normalize = normalize;
                        fin.push_back(f);
                        fresh[de] = nb;
                    }
                }
                if (!fin.empty()) {
                    const MsgFinalItem* d = upload_small(s, fin);
                    ProfScope ps(s, TNQS_PROF_SMALL, 0, 0);
                    launch_msg_finalize<T>(s->stream, d, (int)fin.size());
                }
                ht_fin.stop();
                // (sharded) the next sub-batch writes the exchange buffer again: ordered after this finalize by the stream; the host-side
                // all-gather callback is always preceded by a stream synchronisation inside exchange()
                start = end;
            }
        }
        for (size_t t = 0; t < nseq; ++t) if (fresh[plan.seq[t]]) cur[plan.seq[t]] = fresh[plan.seq[t]];
        s->stats.n_bp_sweeps += 1;
        if (compute_error) {
            launch_sum_doubles(s->stream, reinterpret_cast<const double*>(d_diffs->p), (int)nseq, reinterpret_cast<double*>(d_sum->p));
            if (go_optimistic) {
                // the verdict travels to a pinned slot and stays a Check (engine.hpp): the messages are committed and the caller goes on enqueuing meanwhile
                double* slot = reinterpret_cast<double*>(ring_alloc(s, sizeof(double)));       // (may settle older checks -- and throw -- first: nothing is committed yet)
                HIPCHK(hipMemcpyAsync(slot, d_sum->p, sizeof(double), hipMemcpyDeviceToHost, s->stream));
                Check c; c.kind = 1; c.step = s->cur_step; c.iters_done = iter; c.ev = check_event(s);
                HIPCHK(hipEventRecord(c.ev, s->stream));
                const size_t nseq_ = nseq; const int maxiter_ = maxiter;
                c.eval = [slot, tol, nseq_, iter, maxiter_](State* st) {
                    const double a = *slot / (double)nseq_;
                    st->stats.last_bp_diff = a;
                    if (a <= tol) return true;
                    if (iter >= maxiter_) { st->stats.bp_not_converged += 1; return true; }      // the reference stops here too (and warns)
                    return false;
                };
                s->keepalive.push_back(d_diffs); s->keepalive.push_back(d_sum);
                s->checks.push_back(std::move(c));
                s->msg = cur; s->stats.n_bp_updates += 1;
                soft_sync(s);
                return;
            }
            const double* st_tot = readback<double>(s, d_sum->p, 1);
            sync(s);
            const double tot = *st_tot;                                  // (the arena's memory is untouched until the next upload)
            settle(s, true);                                             // (the stream is drained: whatever was pending has fired; a failed check unwinds this update, s->msg is untouched)
            avg = tot / (double)nseq;
            if (avg <= tol) { converged = true; niter = iter; break; }
        }
    }
    sync(s);
    settle(s, true);
    s->msg = cur;
    if (iters_before == 0) s->stats.n_bp_updates += 1;
    if (compute_error && !converged) s->stats.bp_not_converged += 1;
    s->stats.last_bp_diff = avg;
    if (niter_out) *niter_out = niter;
    if (diff_out) *diff_out = compute_error ? avg : -1.0;
}

void bp_update(State* s, const tnqs_bp_opts* o, int* niter, double* diff) {
    if (s->dtype == TNQS_C64) bp_update_t<float>(s, o, niter, diff); else bp_update_t<double>(s, o, niter, diff);
}

template void bp_update_t<float>(State*, const tnqs_bp_opts*, int*, double*, bool, int);
template void bp_update_t<double>(State*, const tnqs_bp_opts*, int*, double*, bool, int);

}  // namespace tnqs

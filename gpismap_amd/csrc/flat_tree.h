// Host spatial index of the map: point quadtree (DIM = 2) / octree (DIM = 3) with an
// auto-growing root, stored as flat index-based arrays (node pool + point pool) so
// that cluster point lists and the cluster table can be handed to the device as
// plain id arrays.  Behaviour follows reference cpp/src/octree.cpp and
// cpp/src/quadtree.cpp: same strict/inclusive box tests, same child visiting order
// NW(F), NE(F), SW(F), SE(F), [NWB, NEB, SWB, SEB] (octree.cpp:793-801), same
// growth, subdivision and pruning rules -- the order fixes the row order of every
// cluster's kernel matrix.
#pragma once
#include <cmath>
#include <cstdint>
#include <unordered_set>
#include <vector>

namespace gpis {

struct FlatTreeParam {
    float init_half, min_half, min_half_sq, max_half, cluster_half;
    double cluster_eps;      // |half - cluster| tolerance on insert (octree.cpp:325 / quadtree.cpp:238)
    double qleaf_eps_plain;  // octree.cpp:837
    double qleaf_eps_dist;   // octree.cpp:870
    double qdesc_eps;        // octree.cpp:842,875
};

template <int DIM>
struct FlatPoint {  // reference Node / Node3, strct.h:69-129
    float pos[DIM];
    float grad[DIM];
    float val, sigx, sigg;
    int type;
    bool alive;
};

// The cluster cells one insert() reports (the cell on the new point's path, and the one a displaced point lands in):
// a couple of ids at most -- kept inline, no allocation per inserted point.
struct SmallIdSet {
    int v[6];
    int n = 0;
    std::vector<int> more;
    void insert(int x) {
        for (int i = 0; i < n; ++i) if (v[i] == x) return;
        for (int y : more) if (y == x) return;
        if (n < 6) v[n++] = x; else more.push_back(x);
    }
    bool empty() const { return n == 0; }
    template <class F> void for_each(F f) const { for (int i = 0; i < n; ++i) f(v[i]); for (int y : more) f(y); }
};

template <int DIM>
class FlatTree {
public:
    static constexpr int NC = 1 << DIM;
    struct TNode {
        float c[DIM];
        float h, hsq;
        float lo[DIM], hi[DIM];
        int ch[NC];
        int par;
        int pt;        // point id or -1
        int model;     // OnGPIS store slot or -1
        bool leaf, maxDepth, rootLimit, alive;
    };
    using Set = std::unordered_set<int>;
    using InsSet = SmallIdSet;

    explicit FlatTree(const FlatTreeParam& p) : prm(p) {}

    FlatTreeParam prm;
    std::vector<TNode> nodes;
    std::vector<FlatPoint<DIM>> pts;
    std::vector<int> free_nodes, free_pts;
    std::vector<int> pending_free_pts;  // ids freed during the current update (not reused until recycle())
    std::vector<int> released_models;  // model slots of pruned cluster cells (owner drains this)
    int root = -1;
    // Last cluster-level cell a point walk went through: consecutive pixels / map points mostly fall into the same cell,
    // and a walk that starts there gives the same answer as one from the root whenever the point lies inside the cell by
    // more than the rounding of the boxes (every ancestor then contains it and is an inner node without a point of its own).
    mutable int last_cell = -1;

    bool empty() const { return root < 0; }
    void clear() { nodes.clear(); pts.clear(); free_nodes.clear(); free_pts.clear(); pending_free_pts.clear(); released_models.clear(); root = -1; last_cell = -1; }
    // Point ids freed during an update are only recycled at the next one, so an id names one
    // point object for the whole update (the batched ObsGP results are keyed by id).
    void recycle() { free_pts.insert(free_pts.end(), pending_free_pts.begin(), pending_free_pts.end()); pending_free_pts.clear(); }

    int new_point(const float* pos) {
        int id;
        if (!free_pts.empty()) { id = free_pts.back(); free_pts.pop_back(); }
        else { id = (int)pts.size(); pts.emplace_back(); }
        FlatPoint<DIM>& p = pts[id];
        for (int d = 0; d < DIM; ++d) { p.pos[d] = pos[d]; p.grad[d] = 0.f; }
        p.val = 0.f; p.sigx = 0.f; p.sigg = 0.f; p.type = 0; p.alive = true;
        return id;
    }
    // A point object dies when no leaf refers to it any more.
    void drop_point(int id) { if (id >= 0 && pts[id].alive) { pts[id].alive = false; pending_free_pts.push_back(id); } }

    void make_root(const float* c) {  // octree.cpp:33-51: limit flags are not evaluated for the first root
        root = alloc_node(c, prm.init_half, -1);
        nodes[root].maxDepth = false; nodes[root].rootLimit = false;
    }

    bool is_root(int n) const { return nodes[n].par < 0; }
    int get_root(int n) const { while (nodes[n].par >= 0) n = nodes[n].par; return n; }
    bool empty_leaf(int n) const { return nodes[n].leaf && nodes[n].pt < 0; }

    // Where a walk for point p may start: the cached cell if p is well inside it, else the root.
    int walk_start(const float* p) const { return walk_start_from(last_cell, p); }
    int walk_start_from(int n, const float* p) const {
        if (n < 0 || !nodes[n].alive) return root;
        const TNode& t = nodes[n];
        for (int d = 0; d < DIM; ++d) {
            const float tol = 1e-5f * (std::fabs(t.c[d]) + t.h);
            if (!(p[d] > t.lo[d] + tol && p[d] < t.hi[d] - tol)) return root;
        }
        return n;
    }
    bool is_not_new_cached(const float* p) const { return is_not_new(walk_start(p), p); }
    template <class S>
    bool insert_cached(int pid, S* quads) { return insert(walk_start(pts[pid].pos), pid, quads); }

    // octree.cpp:214-293 / :295-411.  quads == nullptr selects the set-less variant.
    template <class S>
    bool insert(int n, int pid, S* quads) {
        const float* p = pts[pid].pos;
        if (!contains(n, p)) {
            if (nodes[n].par < 0) return nodes[n].rootLimit ? false : insert_to_parent(n, pid);
            return false;
        }
        if (nodes[n].maxDepth) {
            if (nodes[n].pt < 0) {
                nodes[n].pt = pid;
                if (DIM == 2 && quads && at_cluster(n)) quads->insert(n);  // quadtree.cpp:227-229
                return true;
            }
            return false;
        }
        if (nodes[n].leaf) {
            if (nodes[n].h > prm.cluster_half) subdivide(n, -1);
            else {
                if (nodes[n].pt < 0) {
                    nodes[n].pt = pid;
                    if (quads && at_cluster(n)) quads->insert(n);
                    return true;
                }
                int old = nodes[n].pt;
                if (sqdist(pts[old].pos, p) < prm.min_half_sq) return false;
                subdivide(n, -1);
                bool kept = false;
                const float* po = pts[old].pos;
                const int oo = only_child(n, po);
                for (int i = (oo < 0 ? 0 : oo); i < (oo < 0 ? NC : oo + 1); ++i) if (insert(nodes[n].ch[i], old, quads)) { kept = true; break; }
                if (!kept) drop_point(old);  // on a child boundary: the reference loses it too
                nodes[n].pt = -1;
            }
        }
        const int only = only_child(n, p);
        for (int i = (only < 0 ? 0 : only); i < (only < 0 ? NC : only + 1); ++i)
            if (insert(nodes[n].ch[i], pid, quads)) {
                if (quads && at_cluster(n)) quads->insert(n);
                return true;
            }
        return false;
    }

    bool is_not_new(int n, const float* p) const { return is_not_new_walk(n, p, &last_cell, nullptr); }  // octree.cpp:431-460
    // cell: where the last cluster-level cell of the walk is noted (the caller's cache); witness: if non-null and the answer
    // is true, the node whose stored point lies within the minimum distance of p.
    bool is_not_new_walk(int n, const float* p, int* cell, int* witness) const {
        if (!contains(n, p)) return false;
        for (;;) {
            if (at_cluster(n)) *cell = n;
            if (empty_leaf(n)) return false;
            if (nodes[n].pt >= 0 && sqdist(pts[nodes[n].pt].pos, p) < prm.min_half_sq) { if (witness) *witness = n; return true; }
            if (nodes[n].leaf) return false;
            const int i = only_child(n, p);
            if (i < 0) {  // next to a splitting plane: the reference's loop over all children
                for (int k = 0; k < NC; ++k) if (is_not_new_walk(nodes[n].ch[k], p, cell, witness)) return true;
                return false;
            }
            n = nodes[n].ch[i];
            if (!contains(n, p)) return false;
        }
    }
    // is_not_new() for MANY points against the tree as it stands, from several threads (nothing is written but the caller's
    // own cache `cell`): answer true comes with its witness (node, point id).  A true answer stays true for as long as that
    // node still stores that point: points live in leaves only, a leaf's box and the boxes and splitting planes of its
    // ancestors never change while it is alive, an ancestor never takes a point of its own, point ids are not reused within
    // an update -- so the walk for p still arrives at the witness and finds the same point at the same distance.  (It stops
    // being true when an insert subdivides the witness and its point moves into a child that does not contain p: then
    // nodes[w].pt != id, and the caller asks again.)  A false answer promises nothing: later inserts can make it true.
    bool is_not_new_frozen(const float* p, int* cell, int* wnode, int* wpt) const {
        int w = -1;
        const bool r = is_not_new_walk(walk_start_from(*cell, p), p, cell, &w);
        *wnode = w; *wpt = r ? nodes[w].pt : -1;
        return r;
    }
    bool witness_holds(int wnode, int wpt) const { return wnode >= 0 && nodes[wnode].alive && nodes[wnode].pt == wpt; }

    // octree.cpp:462-508 (set == nullptr: every child visited) / :510-566.  Removes the point
    // stored at position p (distance^2 < 1e-12) and prunes emptied children.
    bool remove(int n, const float* p, Set* set) {
        if (!contains(n, p)) return false;
        if (empty_leaf(n)) return false;
        if (nodes[n].pt >= 0 && (double)sqdist(pts[nodes[n].pt].pos, p) < 1e-12) {
            drop_point(nodes[n].pt);
            nodes[n].pt = -1;
            return true;
        }
        if (nodes[n].leaf) return false;
        bool res = false;
        const int only = only_child(n, p);
        for (int i = (only < 0 ? 0 : only); i < (only < 0 ? NC : only + 1); ++i) {
            if (set) { if (!res) res |= remove(nodes[n].ch[i], p, set); }
            else res |= remove(nodes[n].ch[i], p, nullptr);
        }
        if (res) {
            bool all = true;
            for (int i = 0; i < NC; ++i) all = all && empty_leaf(nodes[n].ch[i]);
            if (all) {
                for (int i = 0; i < NC; ++i) {
                    if (set) set->erase(nodes[n].ch[i]);
                    free_subtree(nodes[n].ch[i]);
                    nodes[n].ch[i] = -1;
                }
                nodes[n].leaf = true;
            }
        }
        return res;
    }

    // remove() started at the cached cluster cell when p lies well inside it (walk_start): every ancestor of that cell
    // contains p away from its splitting planes, so the root walk would have descended exactly this chain through
    // only_child(); what the recursion does on its way back up -- collapse a node whose children are all empty leaves --
    // is done for the ancestors explicitly.  Same return value, same tree, same set.
    bool remove_cached(const float* p, Set* set) {
        const int c = walk_start(p);
        if (c == root || c < 0) return remove(root, p, set);
        if (!remove(c, p, set)) return false;
        // (an ancestor can only collapse when its child on the path has just become an empty leaf: the walk stops at the first
        // one that has not -- every ancestor above it keeps an inner node among its children)
        for (int below = c, a = nodes[c].par; a >= 0 && empty_leaf(below); below = a, a = nodes[a].par) {
            bool all = true;
            for (int i = 0; i < NC && all; ++i) all = empty_leaf(nodes[a].ch[i]);
            if (!all) break;
            for (int i = 0; i < NC; ++i) {
                if (set) set->erase(nodes[a].ch[i]);
                free_subtree(nodes[a].ch[i]);
                nodes[a].ch[i] = -1;
            }
            nodes[a].leaf = true;
            last_cell = -1;          // (the cached cell was one of the freed children or lies below one)
        }
        return true;
    }

    void query_range(int n, const float* c, float h, std::vector<int>& out) const {  // octree.cpp:777-804
        float lo[DIM], hi[DIM];
        for (int d = 0; d < DIM; ++d) { lo[d] = c[d] - h; hi[d] = c[d] + h; }
        query_range_rec(n, c, h * h, lo, hi, out);
    }

    // Range queries of a batch (updateGPs asks one per cluster, radii overlapping 27 cells each): the points of a cell are
    // listed once (in traversal order) and every query filters the lists of the cells its box touches, visited in
    // traversal order too -- the same points in the same order as query_range(), without walking the subtrees again.
    // (Leaves above the cluster level hold no points, so the cell walk misses nothing.)  Valid while the tree is not modified.
    struct CellLists {
        std::vector<int> begin, end;   // by node id; begin < 0: not listed yet
        std::vector<int> pts, cells;
        void reset(size_t nnodes) { begin.assign(nnodes, -1); end.assign(nnodes, -1); pts.clear(); }
    };
    void query_range_cells(const float* c, float h, CellLists& cl, std::vector<int>& out) const {
        cl.cells.clear();
        query_clusters(root, c, h, cl.cells, nullptr);
        const float hsq = h * h;
        for (int cell : cl.cells) {
            if (cl.begin[cell] < 0) {
                cl.begin[cell] = (int)cl.pts.size();
                all_points(cell, cl.pts);
                cl.end[cell] = (int)cl.pts.size();
            }
            for (int i = cl.begin[cell]; i < cl.end[cell]; ++i) {
                const int pid = cl.pts[i];
                if (sqdist(pts[pid].pos, c) < hsq) out.push_back(pid);
            }
        }
    }

    // The cell part of query_range_cells() alone, for the device-side filter (ongpis_range_gather_kernel): appends one
    // (begin, end) pair into cl.pts per non-empty touched cell, in traversal order; returns the number of listed points.
    int range_cells(const float* c, float h, CellLists& cl, std::vector<int>& ranges) const {
        cl.cells.clear();
        query_clusters(root, c, h, cl.cells, nullptr);
        return range_cells_from(cl.cells, cl, ranges);
    }
    // ... with the cell walk done by the caller (the walks of a frame's clusters are independent and read-only: the map runs
    // them on its host threads; the listing below stays in cluster order, so every offset is what the serial version gives)
    int range_cells_from(const std::vector<int>& cells, CellLists& cl, std::vector<int>& ranges) const {
        int total = 0;
        for (int cell : cells) {
            if (cl.begin[cell] < 0) {
                cl.begin[cell] = (int)cl.pts.size();
                all_points(cell, cl.pts);
                cl.end[cell] = (int)cl.pts.size();
            }
            if (cl.end[cell] > cl.begin[cell]) { ranges.push_back(cl.begin[cell]); ranges.push_back(cl.end[cell]); total += cl.end[cell] - cl.begin[cell]; }
        }
        return total;
    }

    void all_points(int n, std::vector<int>& out) const {  // octree.cpp:806-827
        if (empty_leaf(n)) return;
        if (nodes[n].leaf) { out.push_back(nodes[n].pt); return; }
        for (int i = 0; i < NC; ++i) all_points(nodes[n].ch[i], out);
    }

    // octree.cpp:829-859 (sq == nullptr) / :861-893
    void query_clusters(int n, const float* c, float h, std::vector<int>& out, std::vector<float>* sq) const {
        float lo[DIM], hi[DIM];
        for (int d = 0; d < DIM; ++d) { lo[d] = c[d] - h; hi[d] = c[d] + h; }
        query_clusters_rec(n, c, lo, hi, out, sq);
    }
    void all_clusters(std::vector<int>& out) const {
        if (root < 0) return;
        all_clusters_rec(root, out);
    }

private:
    static float sqdist(const float* a, const float* b) {  // octree.cpp:24-31
        float s = 0.f;
        for (int d = 0; d < DIM; ++d) { float t = a[d] - b[d]; s = (d == 0) ? t * t : s + t * t; }
        return s;
    }
    bool contains(int n, const float* p) const {  // strict, octree.h:119-126
        const TNode& t = nodes[n];
        for (int d = 0; d < DIM; ++d) if (!(p[d] > t.lo[d] && p[d] < t.hi[d])) return false;
        return true;
    }
    bool intersects(int n, const float* lo, const float* hi) const {  // inclusive, octree.h:128-135
        const TNode& t = nodes[n];
        for (int d = 0; d < DIM; ++d) if (hi[d] < t.lo[d] || lo[d] > t.hi[d]) return false;
        return true;
    }
    // The one child whose box can contain p, or -1 when p lies so close to a splitting plane of node n that the rounding
    // of the child boxes (centre -/+ l -/+ l in float) could matter: then the caller visits all children in the reference's
    // order.  Every child-visiting routine above fails immediately on a child that does not (strictly) contain p, so going
    // straight to the only candidate gives the same result and side effects as the reference's loop -- one node touched
    // per level instead of 2^DIM.
    int only_child(int n, const float* p) const {
        const TNode& t = nodes[n];
        int i = 0;
        for (int d = 0; d < DIM; ++d) {
            const float dd = p[d] - t.c[d];
            const float tol = 1e-5f * (std::fabs(t.c[d]) + t.h);
            if (!(std::fabs(dd) > tol)) return -1;
            // child_center(): bit 0 set = +x; bit 1 set = -y; bit 2 set = -z
            if (d == 0) { if (dd > 0.f) i |= 1; }
            else if (dd < 0.f) i |= (1 << d);
        }
        return i;
    }
    // Bit i set = child i of node n certainly fails intersects(lo, hi): its box lies entirely on the far side of one of
    // n's splitting planes, by more than the rounding of the child boxes.  The range walks skip those children without
    // touching their nodes; every child that is visited still gets the reference's exact inclusive test.
    int children_outside(int n, const float* lo, const float* hi) const {
        const TNode& t = nodes[n];
        int skip = 0;
        for (int d = 0; d < DIM; ++d) {
            const float tol = 1e-5f * (std::fabs(t.c[d]) + t.h);
            // child_center(): bit 0 set = +x; bit 1 set = -y; bit 2 set = -z
            const int plus_mask = (d == 0) ? 0xAA : (d == 1 ? 0x33 : 0x0F);   // children on the + side of axis d
            const int all = (1 << NC) - 1;
            if (hi[d] < t.c[d] - tol) skip |= plus_mask & all;             // range entirely below the plane: + children out
            else if (lo[d] > t.c[d] + tol) skip |= (~plus_mask) & all;     // entirely above: - children out
        }
        return skip;
    }
    bool at_cluster(int n) const { return std::fabs((double)(nodes[n].h - prm.cluster_half)) < prm.cluster_eps; }

    int alloc_node(const float* c, float h, int par) {
        int id;
        if (!free_nodes.empty()) { id = free_nodes.back(); free_nodes.pop_back(); }
        else { id = (int)nodes.size(); nodes.emplace_back(); }
        TNode& t = nodes[id];
        t.h = h; t.hsq = h * h;
        for (int d = 0; d < DIM; ++d) { t.c[d] = c[d]; t.lo[d] = c[d] - h; t.hi[d] = c[d] + h; }
        for (int i = 0; i < NC; ++i) t.ch[i] = -1;
        t.par = par; t.pt = -1; t.model = -1;
        t.leaf = true; t.alive = true;
        t.maxDepth = h < prm.min_half;
        t.rootLimit = h > prm.max_half;
        return id;
    }
    void free_subtree(int n) {
        TNode& t = nodes[n];
        if (!t.leaf) for (int i = 0; i < NC; ++i) if (t.ch[i] >= 0) free_subtree(t.ch[i]);
        if (t.pt >= 0) drop_point(t.pt);
        if (t.model >= 0) released_models.push_back(t.model);
        nodes[n].alive = false; nodes[n].model = -1; nodes[n].pt = -1;
        if (n == last_cell) last_cell = -1;
        free_nodes.push_back(n);
    }
    void child_center(int n, int i, float l, float* c) const {
        const TNode& t = nodes[n];
        c[0] = (i & 1) ? t.c[0] + l : t.c[0] - l;
        c[1] = (i & 2) ? t.c[1] - l : t.c[1] + l;
        if (DIM == 3) c[DIM - 1] = (i & 4) ? t.c[DIM - 1] - l : t.c[DIM - 1] + l;
    }
    void subdivide(int n, int except) {  // octree.cpp:670-712 / :714-775
        float l = (float)((double)nodes[n].h * 0.5);
        for (int i = 0; i < NC; ++i) {
            if (i == except) continue;
            float c[DIM];
            child_center(n, i, l, c);
            int id = alloc_node(c, l, n);  // may reallocate `nodes`
            nodes[n].ch[i] = id;
        }
        nodes[n].leaf = false;
    }
    bool insert_to_parent(int n, int pid) {  // octree.cpp:151-212
        const float* np = pts[pid].pos;
        float l = nodes[n].h;
        float pc[DIM];
        for (int d = 0; d < DIM; ++d) pc[d] = 0.f;
        bool strict = true;
        for (int d = 0; d < DIM; ++d) if (!(np[d] > nodes[n].c[d]) && !(np[d] < nodes[n].c[d])) strict = false;
        int slot = 0;
        if (strict) {
            bool plus[3] = {false, false, false};
            for (int d = 0; d < DIM; ++d) {
                plus[d] = np[d] > nodes[n].c[d];
                pc[d] = plus[d] ? nodes[n].c[d] + l : nodes[n].c[d] - l;
            }
            slot = (plus[0] ? 0 : 1) | (plus[1] ? 2 : 0) | ((DIM == 3 && plus[2]) ? 4 : 0);
        }
        int p = alloc_node(pc, (float)(2.0 * (double)l), -1);
        if (strict) {
            subdivide(p, slot);
            nodes[p].ch[slot] = n;
        }  // else: reference quirk -- a childless parent centred at the origin; the old tree is orphaned
        nodes[n].par = p;
        last_cell = -1;   // the tree above the cached cell changed (and may have been orphaned: see the quirk above)
        return insert(p, pid, (Set*)nullptr);
    }
    void query_range_rec(int n, const float* c, float hsq, const float* lo, const float* hi, std::vector<int>& out) const {
        if (!intersects(n, lo, hi) || empty_leaf(n)) return;
        if (nodes[n].leaf) {
            if (sqdist(pts[nodes[n].pt].pos, c) < hsq) out.push_back(nodes[n].pt);
            return;
        }
        const int skip = children_outside(n, lo, hi);
        for (int i = 0; i < NC; ++i) if (!((skip >> i) & 1)) query_range_rec(nodes[n].ch[i], c, hsq, lo, hi, out);
    }
    void query_clusters_rec(int n, const float* c, const float* lo, const float* hi, std::vector<int>& out,
                            std::vector<float>* sq) const {
        if (!intersects(n, lo, hi) || empty_leaf(n)) return;
        const TNode& t = nodes[n];
        if (t.leaf && (double)t.h > (double)prm.cluster_half + (sq ? prm.qleaf_eps_dist : prm.qleaf_eps_plain)) return;
        if ((double)t.h > (double)prm.cluster_half + prm.qdesc_eps) {
            const int skip = children_outside(n, lo, hi);
            for (int i = 0; i < NC; ++i) if (!((skip >> i) & 1)) query_clusters_rec(t.ch[i], c, lo, hi, out, sq);
        } else {
            if (sq) sq->push_back(sqdist(t.c, c));
            out.push_back(n);
        }
    }
    void all_clusters_rec(int n, std::vector<int>& out) const {
        if (empty_leaf(n)) return;
        const TNode& t = nodes[n];
        if (t.leaf && (double)t.h > (double)prm.cluster_half + prm.qleaf_eps_dist) return;
        if ((double)t.h > (double)prm.cluster_half + prm.qdesc_eps) {
            for (int i = 0; i < NC; ++i) all_clusters_rec(t.ch[i], out);
        } else out.push_back(n);
    }

};

}  // namespace gpis

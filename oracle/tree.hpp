// ORACLE (test infrastructure only; parity unpinned -- see linalg.hpp).
// Point quadtree / octree with auto-growing root, restated from the semantics of
// reference cpp/src/octree.cpp + cpp/include/octree.h (DIM = 3) and
// cpp/src/quadtree.cpp + cpp/include/quadtree.h (DIM = 2).  One template covers
// both; child order is the reference's NW(F), NE(F), SW(F), SE(F), [NWB..SEB]:
//   child index bit0 -> +x, bit1 -> -y, bit2 -> -z   (octree.cpp:670-712)
#pragma once
#include <cmath>
#include <memory>
#include <unordered_set>
#include <vector>
#include "gp.hpp"

namespace orc {

template <int DIM>
struct MapNode {  // strct.h:69-129 (Node / Node3)
    float pos[DIM];
    float grad[DIM];
    float val = 0.f, sigx = 0.f, sigg = 0.f;
    int type = 0;
    explicit MapNode(const float* p) {
        for (int d = 0; d < DIM; ++d) { pos[d] = p[d]; grad[d] = 0.f; }
    }
};

struct TreeParam {  // strct.h:175-199
    float init_half, min_half, min_half_sq, max_half, cluster_half;
    // the reference compares against double literals, so these stay double
    double cluster_eps;      // |half - cluster| tolerance on insert (octree.cpp:325 / quadtree.cpp:238)
    double qleaf_eps_plain;  // leaf cut-off in QueryNonEmptyLevelC without distances
    double qleaf_eps_dist;   // ... with distances
    double qdesc_eps;        // descend while half > cluster + qdesc_eps
};

template <int DIM>
struct Box {  // AABB / AABB3, octree.h:31-136
    float c[DIM];
    float h = 0.f, hsq = 0.f;
    float lo[DIM], hi[DIM];
    Box() { for (int d = 0; d < DIM; ++d) c[d] = lo[d] = hi[d] = 0.f; }
    Box(const float* c_, float h_) {
        h = h_; hsq = h * h;
        for (int d = 0; d < DIM; ++d) { c[d] = c_[d]; lo[d] = c[d] - h; hi[d] = c[d] + h; }
    }
    bool contains(const float* p) const {  // strict
        for (int d = 0; d < DIM; ++d) if (!(p[d] > lo[d] && p[d] < hi[d])) return false;
        return true;
    }
    bool intersects(const Box& o) const {  // inclusive
        for (int d = 0; d < DIM; ++d) if (o.hi[d] < lo[d] || o.lo[d] > hi[d]) return false;
        return true;
    }
};

template <int DIM>
static inline float sqdist(const float* a, const float* b) {  // octree.cpp:24-31
    float s = 0.f;
    for (int d = 0; d < DIM; ++d) { float t = a[d] - b[d]; s = (d == 0) ? t * t : s + t * t; }
    return s;
}

template <int DIM>
struct Tree {
    static constexpr int NC = 1 << DIM;
    using NodeP = std::shared_ptr<MapNode<DIM>>;
    using Set = std::unordered_set<Tree*>;

    const TreeParam* prm;
    Box<DIM> box;
    NodeP node;
    std::shared_ptr<OnGPIS> gp;
    bool leaf = true, maxDepth = false, rootLimit = false;
    Tree* ch[NC];
    Tree* par = nullptr;

    Tree(const TreeParam* p, const Box<DIM>& b, Tree* parent) : prm(p), box(b), par(parent) {
        for (auto& c : ch) c = nullptr;
        if (box.h < prm->min_half) maxDepth = true;
        if (box.h > prm->max_half) rootLimit = true;
    }
    // root at c with the initial half length (octree.cpp:33-51: no limit flags evaluated)
    static Tree* make_root(const TreeParam* p, const float* c) {
        Tree* t = new Tree(p, Box<DIM>(c, p->init_half), nullptr);
        t->maxDepth = false; t->rootLimit = false;
        return t;
    }
    ~Tree() { for (auto& c : ch) { delete c; c = nullptr; } }

    bool isRoot() const { return par == nullptr; }
    Tree* root() { Tree* p = this; while (p->par) p = p->par; return p; }
    bool emptyLeaf() const { return leaf && !node; }
    bool atCluster() const { return std::fabs((double)(box.h - prm->cluster_half)) < prm->cluster_eps; }

    static void child_center(const Box<DIM>& b, int i, float l, float* c) {
        c[0] = (i & 1) ? b.c[0] + l : b.c[0] - l;
        c[1] = (i & 2) ? b.c[1] - l : b.c[1] + l;
        if (DIM == 3) c[DIM - 1] = (i & 4) ? b.c[DIM - 1] - l : b.c[DIM - 1] + l;
    }
    void subdivide(int except = -1) {  // octree.cpp:670-712 / :714-775
        float l = (float)((double)box.h * 0.5);
        for (int i = 0; i < NC; ++i) {
            if (i == except) continue;
            float c[DIM];
            child_center(box, i, l, c);
            ch[i] = new Tree(prm, Box<DIM>(c, l), this);
        }
        leaf = false;
    }

    // octree.cpp:151-212.  The new parent is reached through the set-less Insert.
    bool insertToParent(const NodeP& n) {
        float l = box.h;
        float pc[DIM];
        for (int d = 0; d < DIM; ++d) pc[d] = 0.f;
        bool strict = true;
        int slot = 0;
        for (int d = 0; d < DIM; ++d) {
            if (n->pos[d] > box.c[d]) { /* + */ }
            else if (n->pos[d] < box.c[d]) { /* - */ }
            else strict = false;
        }
        if (strict) {
            bool plus[3] = {false, false, false};
            for (int d = 0; d < DIM; ++d) {
                plus[d] = n->pos[d] > box.c[d];
                pc[d] = plus[d] ? box.c[d] + l : box.c[d] - l;
            }
            // this node sits on the opposite side of the new centre
            slot = (plus[0] ? 0 : 1) | (plus[1] ? 2 : 0) | ((DIM == 3 && plus[2]) ? 4 : 0);
        }
        Tree* p = new Tree(prm, Box<DIM>(pc, (float)(2.0 * (double)l)), nullptr);
        if (strict) {
            p->subdivide(slot);
            p->ch[slot] = this;
        }  // else: childType 0 -> a childless leaf centred at the origin (reference quirk)
        par = p;
        return p->insert(n, nullptr);
    }

    // octree.cpp:214-293 (no set) and :295-411 (with set); quads == nullptr selects the former.
    bool insert(const NodeP& n, Set* quads) {
        if (!box.contains(n->pos)) {
            if (!par) return rootLimit ? false : insertToParent(n);
            return false;
        }
        if (maxDepth) {
            if (!node) {
                node = n;
                if (DIM == 2 && quads && atCluster()) quads->insert(this);  // quadtree.cpp:227-229
                return true;
            }
            return false;
        }
        if (leaf) {
            if (box.h > prm->cluster_half) subdivide();
            else {
                if (!node) {
                    node = n;
                    if (quads && atCluster()) quads->insert(this);
                    return true;
                }
                if (sqdist<DIM>(node->pos, n->pos) < prm->min_half_sq) return false;
                subdivide();
                for (int i = 0; i < NC; ++i) if (ch[i]->insert(node, quads)) break;
                node = nullptr;
            }
        }
        for (int i = 0; i < NC; ++i)
            if (ch[i]->insert(n, quads)) {
                if (quads && atCluster()) quads->insert(this);
                return true;
            }
        return false;  // (quadtree.cpp:312 falls off the end here; treated as false)
    }

    bool isNotNew(const NodeP& n) const {  // octree.cpp:431-460
        if (!box.contains(n->pos)) return false;
        if (emptyLeaf()) return false;
        if (node && sqdist<DIM>(node->pos, n->pos) < prm->min_half_sq) return true;
        if (leaf) return false;
        for (int i = 0; i < NC; ++i) if (ch[i]->isNotNew(n)) return true;
        return false;
    }

    // octree.cpp:462-508 (set == nullptr: visits every child) / :510-566 (short-circuits,
    // erases pruned children from the set).
    bool remove(const NodeP& n, Set* set) {
        if (!box.contains(n->pos)) return false;
        if (emptyLeaf()) return false;
        if (node && (double)sqdist<DIM>(node->pos, n->pos) < 1e-12) { node = nullptr; return true; }
        if (leaf) return false;
        bool res = false;
        for (int i = 0; i < NC; ++i) {
            if (set) { if (!res) res |= ch[i]->remove(n, set); }
            else res |= ch[i]->remove(n, nullptr);
        }
        if (res) {
            bool all = true;
            for (int i = 0; i < NC; ++i) all = all && ch[i]->emptyLeaf();
            if (all) {
                for (int i = 0; i < NC; ++i) { if (set) set->erase(ch[i]); delete ch[i]; ch[i] = nullptr; }
                leaf = true;
            }
        }
        return res;
    }

    void queryRange(const Box<DIM>& range, std::vector<NodeP>& out) const {  // octree.cpp:777-804
        if (!box.intersects(range) || emptyLeaf()) return;
        if (leaf) {
            if (sqdist<DIM>(node->pos, range.c) < range.hsq) out.push_back(node);
            return;
        }
        for (int i = 0; i < NC; ++i) ch[i]->queryRange(range, out);
    }

    void allNodes(std::vector<NodeP>& out) const {  // octree.cpp:806-827
        if (emptyLeaf()) return;
        if (leaf) { out.push_back(node); return; }
        for (int i = 0; i < NC; ++i) ch[i]->allNodes(out);
    }

    // octree.cpp:829-859 (sq == nullptr) / :861-893
    void queryClusters(const Box<DIM>& range, std::vector<Tree*>& out, std::vector<float>* sq) {
        if (!box.intersects(range) || emptyLeaf()) return;
        if (leaf && (double)box.h > (double)prm->cluster_half + (sq ? prm->qleaf_eps_dist : prm->qleaf_eps_plain)) return;
        if ((double)box.h > (double)prm->cluster_half + prm->qdesc_eps) {
            for (int i = 0; i < NC; ++i) ch[i]->queryClusters(range, out, sq);
        } else {
            if (sq) sq->push_back(sqdist<DIM>(box.c, range.c));
            out.push_back(this);
        }
    }
};

}  // namespace orc

// TEST: the product's flat index-based tree (gpismap_amd/csrc/flat_tree.h) against the oracle's pointer tree
// (oracle/tree.hpp) on random insert / remove / query sequences, including points ON splitting planes and
// near-duplicates.  Every return value and the traversal order of the stored points must agree.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "flat_tree.h"
#include "tree.hpp"

template <int DIM>
static int run(unsigned seed, int nops, float spread, bool quiet) {
    using FT = gpis::FlatTree<DIM>;
    gpis::FlatTreeParam fp;
    orc::TreeParam op;
    if (DIM == 3) { fp.min_half = (float)(0.0125 / 2.0); fp.max_half = 1.6f; fp.init_half = 0.4f; fp.cluster_half = 0.025f; }
    else { fp.min_half = (float)(0.2 / 2.0); fp.max_half = 102.4f; fp.init_half = 12.8f; fp.cluster_half = 0.8f; }
    fp.min_half_sq = fp.min_half * fp.min_half;
    fp.cluster_eps = 1e-6; fp.qleaf_eps_plain = 0.0001; fp.qleaf_eps_dist = 0.001; fp.qdesc_eps = 0.001;
    op.init_half = fp.init_half; op.min_half = fp.min_half; op.min_half_sq = fp.min_half_sq; op.max_half = fp.max_half;
    op.cluster_half = fp.cluster_half; op.cluster_eps = fp.cluster_eps; op.qleaf_eps_plain = fp.qleaf_eps_plain;
    op.qleaf_eps_dist = fp.qleaf_eps_dist; op.qdesc_eps = fp.qdesc_eps;

    FT ft(fp);
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    float c0[3] = {0.f, 0.f, 0.f};
    ft.make_root(c0);
    orc::Tree<DIM>* ot = orc::Tree<DIM>::make_root(&op, c0);
    std::vector<std::vector<float>> live;
    const float cell = 2.f * fp.cluster_half;
    int bad = 0, inserted = 0, removed = 0, onplane = 0;
    float prev[3] = {0, 0, 0};
    bool have_prev = false;
    // probes of the frozen pre-pass (evalPoints): "not new" answers taken at one moment with their witnesses; while a witness
    // holds, the answer must still be what a fresh walk says, whatever was inserted or removed since
    struct Probe { float p[3]; int wnode, wpt; };
    std::vector<Probe> probes;
    int probes_held = 0, probes_lost = 0;
    for (int it = 0; it < nops && !bad; ++it) {
        if (it % 400 == 0 && !live.empty()) {
            probes.clear();
            int cellc = -1;
            for (int k = 0; k < 96; ++k) {
                Probe pr{};
                const auto& q = live[rng() % live.size()];
                for (int d = 0; d < DIM; ++d) pr.p[d] = q[d] + ((k & 3) == 0 ? 2.0f : 0.6f) * fp.min_half * U(rng);
                const bool r = ft.is_not_new_frozen(pr.p, &cellc, &pr.wnode, &pr.wpt);
                if (r != ft.is_not_new(ft.root, pr.p)) { fprintf(stderr, "op %d: frozen and root walks disagree\n", it); ++bad; }
                if (r) probes.push_back(pr);
            }
        } else if (it % 8 == 0) {
            for (const Probe& pr : probes) {
                if (ft.witness_holds(pr.wnode, pr.wpt)) {
                    ++probes_held;
                    if (!ft.is_not_new(ft.root, pr.p)) { fprintf(stderr, "op %d: witness holds but the walk says new\n", it); ++bad; }
                } else ++probes_lost;
            }
        }
        int kind = rng() % 10;
        if (kind < 7 || live.empty()) {
            float p[3] = {0, 0, 0};
            int mode = rng() % 8;
            for (int d = 0; d < DIM; ++d) p[d] = spread * U(rng);
            if (mode >= 5 && have_prev) {        // a neighbour of the previous candidate (scan-line coherence: exercises the cell cache)
                for (int d = 0; d < DIM; ++d) p[d] = prev[d] + 0.3f * cell * U(rng);
            }
            if (mode == 0 && !live.empty()) {  // near-duplicate of a stored point
                const auto& q = live[rng() % live.size()];
                for (int d = 0; d < DIM; ++d) p[d] = q[d] + 0.7f * fp.min_half * U(rng);
            } else if (mode == 1) {            // on a splitting plane in one or more axes
                for (int d = 0; d < DIM; ++d) if (rng() & 1) p[d] = cell * (float)((int)(p[d] / cell)) * ((rng() & 1) ? 1.f : 0.5f);
                ++onplane;
            } else if (mode == 2) {            // a hair off a plane
                int d = rng() % DIM;
                p[d] = std::nextafterf(cell * (float)((int)(p[d] / cell)), (rng() & 1) ? 1e9f : -1e9f);
            }
            for (int d = 0; d < DIM; ++d) prev[d] = p[d];
            have_prev = true;
            // the product's try_insert
            int pid = ft.new_point(p);
            typename FT::InsSet ins;
            // (the walks start at the cached cluster cell when the point is well inside it -- as the product's try_insert does)
            bool f_notnew = ft.is_not_new_cached(p), f_ok = false;
            if (f_notnew != ft.is_not_new(ft.root, p)) { fprintf(stderr, "op %d: cached and root walks disagree\n", it); ++bad; }
            if (!f_notnew) {
                f_ok = ft.insert_cached(pid, &ins);
                if (f_ok && !ft.is_root(ft.root)) ft.root = ft.get_root(ft.root);
            }
            if (!f_ok) ft.drop_point(pid);
            // the oracle's
            auto n = std::make_shared<orc::MapNode<DIM>>(p);
            typename orc::Tree<DIM>::Set oset;
            bool o_notnew = ot->isNotNew(n), o_ok = false;
            if (!o_notnew) {
                o_ok = ot->insert(n, &oset);
                if (o_ok && !ot->isRoot()) ot = ot->root();
            }
            int fcount = 0; ins.for_each([&](int) { ++fcount; });
            if (f_notnew != o_notnew || f_ok != o_ok || fcount != (int)oset.size()) {
                fprintf(stderr, "op %d insert mismatch: notnew %d/%d ok %d/%d set %d/%zu\n", it, f_notnew, o_notnew, f_ok, o_ok, fcount, oset.size());
                ++bad;
            }
            if (o_ok) { live.push_back(std::vector<float>(p, p + DIM)); ++inserted; }
        } else if (kind < 9) {
            size_t k = rng() % live.size();
            float p[3] = {0, 0, 0};
            for (int d = 0; d < DIM; ++d) p[d] = live[k][d];
            typename FT::Set fset;
            typename orc::Tree<DIM>::Set oset;
            bool with_set = rng() & 1;
            // half of the removals through the cached-cell walk the map's replay uses (primed by a lookup of the same point,
            // as the replay's previous insert does), half from the root
            const bool cached = (rng() & 1) != 0;
            if (cached && (rng() & 1)) (void)ft.is_not_new(ft.root, p);
            bool fr = cached ? ft.remove_cached(p, with_set ? &fset : nullptr) : ft.remove(ft.root, p, with_set ? &fset : nullptr);
            auto n = std::make_shared<orc::MapNode<DIM>>(p);
            bool orr = ot->remove(n, with_set ? &oset : nullptr);
            if (fr != orr) { fprintf(stderr, "op %d remove mismatch %d/%d\n", it, fr, orr); ++bad; }
            live[k] = live.back(); live.pop_back();
            ++removed;
        } else {
            float c[3] = {0, 0, 0};
            for (int d = 0; d < DIM; ++d) c[d] = spread * U(rng);
            float h = cell * (0.5f + 2.f * std::fabs(U(rng)));
            std::vector<int> fr;
            ft.query_range(ft.root, c, h, fr);
            std::vector<typename orc::Tree<DIM>::NodeP> orr;
            ot->queryRange(orc::Box<DIM>(c, h), orr);
            bool same = fr.size() == orr.size();
            for (size_t i = 0; same && i < fr.size(); ++i)
                for (int d = 0; d < DIM; ++d) same = same && ft.pts[fr[i]].pos[d] == orr[i]->pos[d];
            if (!same) { fprintf(stderr, "op %d range query mismatch (%zu / %zu)\n", it, fr.size(), orr.size()); ++bad; }
            // the batched form updateGPs uses: cell lists + filter
            {
                typename FT::CellLists cl;
                cl.reset(ft.nodes.size());
                for (int rep = 0; rep < 2; ++rep) {   // second pass: the lists are reused
                    std::vector<int> fr2;
                    ft.query_range_cells(c, h, cl, fr2);
                    if (fr2 != fr) { fprintf(stderr, "op %d cell-list range query mismatch (%zu / %zu)\n", it, fr2.size(), fr.size()); ++bad; }
                }
            }
            // cluster-cell query, with and without distances (octree.cpp:829-893)
            for (int with_sq = 0; with_sq < 2; ++with_sq) {
                std::vector<int> fc;
                std::vector<float> fsq, osq;
                ft.query_clusters(ft.root, c, h, fc, with_sq ? &fsq : nullptr);
                std::vector<orc::Tree<DIM>*> oc;
                ot->queryClusters(orc::Box<DIM>(c, h), oc, with_sq ? &osq : nullptr);
                bool ok = fc.size() == oc.size() && fsq.size() == osq.size();
                for (size_t i = 0; ok && i < fc.size(); ++i) {
                    for (int d = 0; d < DIM; ++d) ok = ok && ft.nodes[fc[i]].c[d] == oc[i]->box.c[d];
                    if (with_sq) ok = ok && fsq[i] == osq[i];
                }
                if (!ok) { fprintf(stderr, "op %d cluster query mismatch (%zu / %zu)\n", it, fc.size(), oc.size()); ++bad; }
            }
        }
        if (it % 257 == 0 || it == nops - 1) {
            std::vector<int> fa;
            ft.all_points(ft.root, fa);
            std::vector<typename orc::Tree<DIM>::NodeP> oa;
            ot->allNodes(oa);
            bool same = fa.size() == oa.size();
            for (size_t i = 0; same && i < fa.size(); ++i)
                for (int d = 0; d < DIM; ++d) same = same && ft.pts[fa[i]].pos[d] == oa[i]->pos[d];
            if (!same) { fprintf(stderr, "op %d traversal mismatch (%zu / %zu points)\n", it, fa.size(), oa.size()); ++bad; }
        }
    }
    std::vector<int> fa;
    ft.all_points(ft.root, fa);
    if (!quiet) printf("dim %d seed %u: frozen probes checked while their witness held %d, witnesses lost %d\n", DIM, seed, probes_held, probes_lost);
    if (!quiet) printf("dim %d seed %u: %d ops, %d inserted, %d removed, %d on-plane candidates, %zu stored, %s\n", DIM, seed, nops, inserted,
                       removed, onplane, fa.size(), bad ? "MISMATCH" : "identical");
    delete ot;
    return bad;
}

int main(int argc, char** argv) {
    int nops = argc > 1 ? atoi(argv[1]) : 40000;
    int bad = 0;
    for (unsigned seed = 1; seed <= 3; ++seed) {
        bad += run<3>(seed, nops, 0.35f, false);
        bad += run<3>(seed + 10, nops, 2.5f, false);   // outgrows the root several times
        bad += run<2>(seed, nops, 9.f, false);
        bad += run<2>(seed + 10, nops, 60.f, false);
    }
    return bad ? 1 : 0;
}

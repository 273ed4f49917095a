// Exact emulation of GNU libstdc++ std::sort -- usable from host (unit test) and device code.
#pragma once
#if defined(__HIPCC__)
#define GPIS_HD __host__ __device__
#else
#define GPIS_HD
#endif

namespace gpis {

// Exact emulation of GNU libstdc++ std::sort (introsort: median-of-3 unguarded partition while
// a range is longer than 16, then insertion sort) on an index array ordered by key[idx].  The
// reference sorts candidate cells by squared centre distance with std::sort
// (GPisMap3.cpp:826-829); with exact distance ties (lattice-aligned queries) the outcome depends
// on this algorithm, so the tie path reproduces it operation by operation.
// key / v / the explicit recursion stack are array VIEWS (anything with operator[]: plain pointers on the host and in
// private memory, Strided<> over a lane-interleaved LDS block in the device tie kernel).
template <class T>
struct Strided {
    T* p; int stride;
    GPIS_HD T& operator[](int i) const { return p[(size_t)i * stride]; }
};
#define CMP(a, b) (key[(a)] < key[(b)])
template <class KA, class VA>
GPIS_HD inline void unguarded_linear_insert(KA key, VA v, int last) {
    int val = v[last];
    int next = last - 1;
    while (CMP(val, v[next])) { v[last] = v[next]; last = next; --next; }
    v[last] = val;
}
template <class KA, class VA>
GPIS_HD inline void insertion_sort(KA key, VA v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (CMP(v[i], v[first])) {
            int val = v[i];
            for (int k = i; k > first; --k) v[k] = v[k - 1];
            v[first] = val;
        } else unguarded_linear_insert(key, v, i);
    }
}
template <class KA, class VA, class SA>
GPIS_HD inline bool stdsort_emulate(KA key, VA v, int n, SA stF, SA stL, SA stD) {
    if (n <= 1) return true;
    int depth = 0;
    for (int t = n; t > 1; t >>= 1) ++depth;
    depth *= 2;
    int sp = 0;
    stF[0] = 0; stL[0] = n; stD[0] = depth; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stF[sp], last = stL[sp], dl = stD[sp];
        while (last - first > 16) {
            if (dl == 0) return false;  // heapsort fallback of libstdc++: not emulated (never reached for n <= 128)
            --dl;
            int mid = first + (last - first) / 2;
            int a = first + 1, b = mid, c = last - 1;
            // __move_median_to_first(first, a, b, c)
            int sel;
            if (CMP(v[a], v[b])) { if (CMP(v[b], v[c])) sel = b; else if (CMP(v[a], v[c])) sel = c; else sel = a; }
            else if (CMP(v[a], v[c])) sel = a; else if (CMP(v[b], v[c])) sel = c; else sel = b;
            { int tmp = v[first]; v[first] = v[sel]; v[sel] = tmp; }
            // __unguarded_partition(first+1, last, pivot = first)
            int lo = first + 1, hi = last;
            const int piv = v[first];
            while (true) {
                while (CMP(v[lo], piv)) ++lo;
                --hi;
                while (CMP(piv, v[hi])) --hi;
                if (!(lo < hi)) break;
                int tmp = v[lo]; v[lo] = v[hi]; v[hi] = tmp;
                ++lo;
            }
            int cut = lo;
            if (sp >= 32) return false;
            stF[sp] = cut; stL[sp] = last; stD[sp] = dl; ++sp;  // right part later (the recursion of libstdc++)
            last = cut;
        }
    }
    // NB: libstdc++ recurses into the right part BEFORE continuing with the left one; the
    // partitions are disjoint ranges, so the processing order does not change the result.
    if (n > 16) {
        insertion_sort(key, v, 0, 16);
        for (int i = 16; i < n; ++i) unguarded_linear_insert(key, v, i);
    } else insertion_sort(key, v, 0, n);
    return true;
}
GPIS_HD inline bool stdsort_emulate(const float* key, int* v, int n) {
    int stF[32], stL[32], stD[32];
    return stdsort_emulate<const float*, int*, int*>(key, v, n, stF, stL, stD);
}
#undef CMP

}  // namespace gpis

// TEST-ONLY implementation of the mx* subset declared in this directory's mex.h (see its header).
#include "mex.h"
#include <cstdlib>
#include <cstring>

struct mxArray_tag {
    mxClassID cls;
    mwSize dims[2];
    void* data;
    size_t elem;
};

static mxArray* make(mxClassID cls, mwSize m, mwSize n, size_t elem) {
    mxArray* a = (mxArray*)std::calloc(1, sizeof(mxArray));
    a->cls = cls; a->dims[0] = m; a->dims[1] = n; a->elem = elem;
    a->data = std::calloc(m * n ? m * n : 1, elem);   // MATLAB zero-fills created matrices
    return a;
}

extern "C" {

int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (!a || a->cls != mxCHAR_CLASS || buflen == 0) return 1;
    size_t n = a->dims[0] * a->dims[1];
    int rc = 0;
    if (n > buflen - 1) { n = buflen - 1; rc = 1; }
    std::memcpy(buf, a->data, n);
    buf[n] = 0;
    return rc;
}
mwSize mxGetNumberOfDimensions(const mxArray*) { return 2; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
void* mxGetData(const mxArray* a) { return a->data; }
size_t mxGetNumberOfElements(const mxArray* a) { return a->dims[0] * a->dims[1]; }
double* mxGetPr(const mxArray* a) { return (double*)a->data; }
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return make(mxDOUBLE_CLASS, m, n, sizeof(double)); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) {
    return make(cls, m, n, cls == mxDOUBLE_CLASS ? sizeof(double) : (cls == mxSINGLE_CLASS ? sizeof(float) : 1));
}
void mxDestroyArray(mxArray* a) { if (a) { std::free(a->data); std::free(a); } }

mxArray* mxstub_string(const char* s) {
    size_t n = std::strlen(s);
    mxArray* a = make(mxCHAR_CLASS, 1, n, 1);
    std::memcpy(a->data, s, n);
    return a;
}
mxArray* mxstub_single(const float* d, mwSize m, mwSize n) {
    mxArray* a = make(mxSINGLE_CLASS, m, n, sizeof(float));
    if (d) std::memcpy(a->data, d, m * n * sizeof(float));
    return a;
}
mxArray* mxstub_double(const double* d, mwSize m, mwSize n) {
    mxArray* a = make(mxDOUBLE_CLASS, m, n, sizeof(double));
    if (d) std::memcpy(a->data, d, m * n * sizeof(double));
    return a;
}
int mxstub_call(int nlhs, mxArray* out[2], int nrhs, mxArray* a0, mxArray* a1, mxArray* a2, mxArray* a3) {
    mxArray* plhs[4] = {0, 0, 0, 0};
    const mxArray* prhs[4] = {a0, a1, a2, a3};
    mexFunction(nlhs, plhs, nrhs, prhs);
    int n = 0;
    for (int i = 0; i < 2; ++i) { out[i] = plhs[i]; if (plhs[i]) ++n; }
    return n;
}

}  // extern "C"

/* TEST-ONLY stand-in for MATLAB's <mex.h> (MATLAB is not installed in the build image or on the
 * GPU box).  It declares exactly the part of the mx* API that the reference's two gateways use
 * (reference mex/mexGPisMap3.cpp:49-166, mex/mexGPisMap.cpp:40-131; list in SURVEY.md 8(b)) so that
 * those gateway sources can be compiled UNCHANGED, by path, against gpismap_amd's include/ and driven
 * from a test.  It is not part of the product and is never shipped: a real build uses MATLAB's own
 * mex.h through mex/make_GPisMap3_amd.m / make_GPisMap_amd.m. */
#ifndef GPISMAP_AMD_TEST_MEXSTUB_H_
#define GPISMAP_AMD_TEST_MEXSTUB_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef enum { mxUNKNOWN_CLASS = 0, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef struct mxArray_tag mxArray;

int mxGetString(const mxArray* a, char* buf, mwSize buflen);
mwSize mxGetNumberOfDimensions(const mxArray* a);
const mwSize* mxGetDimensions(const mxArray* a);
mxClassID mxGetClassID(const mxArray* a);
void* mxGetData(const mxArray* a);
size_t mxGetNumberOfElements(const mxArray* a);
double* mxGetPr(const mxArray* a);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);                  /* zero-filled */
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag); /* zero-filled */
void mxDestroyArray(mxArray* a);

/* the gateway entry point (defined by the gateway source under test) */
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

/* ---- helpers for the test driver (not MATLAB API) ---- */
mxArray* mxstub_string(const char* s);
mxArray* mxstub_single(const float* data, mwSize m, mwSize n);   /* copies */
mxArray* mxstub_double(const double* data, mwSize m, mwSize n);  /* copies */
/* call mexFunction with up to 4 right-hand sides and up to 2 left-hand sides; returns the number of
 * outputs the gateway created (plhs[i] != NULL), outputs in out[0..1] (caller frees with mxDestroyArray) */
int mxstub_call(int nlhs, mxArray* out[2], int nrhs, mxArray* a0, mxArray* a1, mxArray* a2, mxArray* a3);

#ifdef __cplusplus
}
#endif
#endif

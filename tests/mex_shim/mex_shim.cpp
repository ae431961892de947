// TEST INFRASTRUCTURE -- a minimal *functional* stand-in for the MEX runtime, NOT MATLAB's libmex / libmx.
//
// The build image has neither MATLAB nor Octave, so mex/cp_mex.cpp had only ever been parsed (g++ -fsyntax-only against
// mex/proto/mex.h). This file defines the MEX C API entry points that header declares -- a plain mxArray record (class,
// up to three dimensions, a malloc'ed column-major payload), mexErrMsgIdAndTxt as MATLAB implements it (it records the
// message id and text and longjmps out of mexFunction), mexLock / mexUnlock / mexAtExit bookkeeping -- plus a small C
// API (mxt_*) through which tests/test_gpu_mex_gateway.py builds argument lists, calls the gateway's real mexFunction and
// reads the outputs. cp_mex.cpp is compiled unchanged and linked against the real libcpengine.so:
//     g++ -std=c++17 -shared -fPIC -I mex/proto -I include mex/cp_mex.cpp tests/mex_shim/mex_shim.cpp -L ma_amd -lcpengine
// Nothing of the product links against this file.
#include "mex.h"

#include <csetjmp>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct mxArray_tag {
    mxClassID cls;
    int is_char;
    mwSize ndim;
    mwSize dims[3];
    void *data;       // column-major payload (char arrays: one byte per character)
    size_t elem;      // bytes per element
};

namespace {
jmp_buf g_jmp;
bool g_in_call = false;
char g_err_id[128], g_err_msg[1024];
int g_lock = 0;
void (*g_atexit)(void) = nullptr;
long g_live_arrays = 0;

size_t numel(const mxArray *a) { return a->dims[0] * a->dims[1] * (a->ndim > 2 ? a->dims[2] : 1); }
size_t elem_size(mxClassID c) { return c == mxDOUBLE_CLASS ? 8 : c == mxUINT64_CLASS ? 8 : c == mxINT32_CLASS ? 4 : 1; }
mxArray *make(mxClassID cls, mwSize m, mwSize n, mwSize p, int is_char) {
    mxArray *a = static_cast<mxArray *>(calloc(1, sizeof(mxArray)));
    a->cls = cls; a->is_char = is_char; a->ndim = p > 1 ? 3 : 2;
    a->dims[0] = m; a->dims[1] = n; a->dims[2] = p > 1 ? p : 1;
    a->elem = elem_size(cls);
    const size_t bytes = numel(a) * a->elem;
    a->data = calloc(bytes ? bytes : 1, 1);    // mxCreate* arrays are zero-initialised
    ++g_live_arrays;
    return a;
}
}   // namespace

extern "C" {
// ---- the MEX C API subset of mex/proto/mex.h --------------------------------------------------------------------------
double *mxGetPr(const mxArray *a) { return static_cast<double *>(a->data); }
void *mxGetData(const mxArray *a) { return a->data; }
double mxGetScalar(const mxArray *a) {
    if (a->cls == mxDOUBLE_CLASS) return *static_cast<double *>(a->data);
    if (a->cls == mxUINT64_CLASS) return (double)*static_cast<uint64_t *>(a->data);
    if (a->cls == mxINT32_CLASS) return (double)*static_cast<int32_t *>(a->data);
    return (double)*static_cast<unsigned char *>(a->data);
}
size_t mxGetNumberOfElements(const mxArray *a) { return numel(a); }
size_t mxGetM(const mxArray *a) { return a->dims[0]; }
size_t mxGetN(const mxArray *a) { return a->dims[1] * (a->ndim > 2 ? a->dims[2] : 1); }   // product of the trailing dimensions
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->ndim; }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims; }
bool mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS && !a->is_char; }
bool mxIsComplex(const mxArray *) { return false; }
bool mxIsEmpty(const mxArray *a) { return numel(a) == 0; }
bool mxIsChar(const mxArray *a) { return a->is_char != 0; }
mxClassID mxGetClassID(const mxArray *a) { return a->cls; }
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {
    const size_t n = numel(a);
    if (!a->is_char || n + 1 > buflen) return 1;
    memcpy(buf, a->data, n);
    buf[n] = 0;
    return 0;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return make(mxDOUBLE_CLASS, m, n, 1, 0); }
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = make(mxDOUBLE_CLASS, 1, 1, 1, 0); *static_cast<double *>(a->data) = v; return a; }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) { return make(cls, m, n, 1, 0); }
mxArray *mxCreateString(const char *s) {
    const size_t n = strlen(s);
    mxArray *a = make(mxUNKNOWN_CLASS, 1, n, 1, 1);
    memcpy(a->data, s, n);
    return a;
}
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    snprintf(g_err_id, sizeof g_err_id, "%s", id ? id : "");
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err_msg, sizeof g_err_msg, fmt, ap);
    va_end(ap);
    if (!g_in_call) { fprintf(stderr, "mex shim: error outside a call: %s %s\n", g_err_id, g_err_msg); abort(); }
    longjmp(g_jmp, 1);     // what MATLAB does: control never returns to the MEX function
}
void mexLock(void) { ++g_lock; }
void mexUnlock(void) { --g_lock; }
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
int mexPrintf(const char *fmt, ...) { va_list ap; va_start(ap, fmt); const int r = vprintf(fmt, ap); va_end(ap); return r; }

// ---- test driver API -----------------------------------------------------------------------------------------------------
mxArray *mxt_double(size_t m, size_t n, size_t p, const double *src) {
    mxArray *a = make(mxDOUBLE_CLASS, m, n, p, 0);
    if (src && numel(a)) memcpy(a->data, src, numel(a) * 8);
    return a;
}
mxArray *mxt_string(const char *s) { return mxCreateString(s); }
mxArray *mxt_uint64(uint64_t v) { mxArray *a = make(mxUINT64_CLASS, 1, 1, 1, 0); *static_cast<uint64_t *>(a->data) = v; return a; }
void mxt_free(mxArray *a) { if (a) { free(a->data); free(a); --g_live_arrays; } }
size_t mxt_numel(const mxArray *a) { return numel(a); }
size_t mxt_dim(const mxArray *a, int d) { return d < (int)a->ndim ? a->dims[d] : 1; }
int mxt_class(const mxArray *a) { return a->is_char ? -1 : (int)a->cls; }
const void *mxt_data(const mxArray *a) { return a->data; }
const char *mxt_error_id(void) { return g_err_id; }
const char *mxt_error_msg(void) { return g_err_msg; }
int mxt_lock_count(void) { return g_lock; }
long mxt_live_arrays(void) { return g_live_arrays; }
void mxt_run_atexit(void) { if (g_atexit) g_atexit(); }
// calls the gateway; 0 = returned normally, 1 = left through mexErrMsgIdAndTxt (id / text in mxt_error_*). plhs must have
// room for max(nlhs, 1) outputs; entries the gateway did not set stay NULL.
int mxt_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {
    for (int i = 0; i < (nlhs > 1 ? nlhs : 1); ++i) plhs[i] = nullptr;
    g_err_id[0] = g_err_msg[0] = 0;
    g_in_call = true;
    if (setjmp(g_jmp) != 0) { g_in_call = false; return 1; }
    mexFunction(nlhs, plhs, nrhs, prhs);
    g_in_call = false;
    return 0;
}
}   // extern "C"

/* PROTOTYPE HEADER -- NOT MATLAB'S mex.h.
 * The build image has neither MATLAB nor Octave, so cp_mex.cpp cannot be compiled against the real MEX API here. This
 * file declares the handful of documented MEX C API entry points the gateway uses (names, argument order and types as
 * in the public MEX reference), for ONE purpose: `g++ -fsyntax-only -I mex/proto mex/cp_mex.cpp`, run by
 * __graft_entry__.build(), so that the gateway at least parses and type-checks. Nothing links against it and it defines
 * nothing. A maintainer builds the gateway with the real header:  mex -I../include cp_mex.cpp -L../ma_amd -lcpengine
 * (MATLAB)  or  mkoctfile --mex -I../include cp_mex.cpp -L../ma_amd -lcpengine  (Octave). */
#ifndef CP_PROTO_MEX_H
#define CP_PROTO_MEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12, mxUINT64_CLASS = 15 } mxClassID;
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
bool mxIsDouble(const mxArray *a);
bool mxIsComplex(const mxArray *a);
bool mxIsEmpty(const mxArray *a);
bool mxIsChar(const mxArray *a);
mxClassID mxGetClassID(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag);
mxArray *mxCreateString(const char *s);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexLock(void);
void mexUnlock(void);
int mexAtExit(void (*fn)(void));
int mexPrintf(const char *fmt, ...);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif

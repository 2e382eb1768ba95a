/* TEST-ONLY declarations of the R C API symbols bmm-mcmc_amd/r-shim/bmmmcmc_shim.c uses, so that
 * tests/test_r_shim.py can run `gcc -fsyntax-only -Wall -Werror` on the shim in an image without R.
 * Signatures as documented in "Writing R Extensions" (sections 5.9, 5.10, 6.3); nothing here is
 * linked, run, or shipped, and it is not a stand-in for building the reference package. */
#ifndef TEST_STUB_RINTERNALS_H
#define TEST_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define RAWSXP 24
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
int R_IsNaN(double);
#define ISNAN(x) (R_IsNaN(x) || (x) != (x))
int TYPEOF(SEXP);
Rboolean Rf_isMatrix(SEXP);
int Rf_nrows(SEXP), Rf_ncols(SEXP);
int Rf_asInteger(SEXP), Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
SEXP Rf_allocVector(unsigned int, R_xlen_t), Rf_allocMatrix(unsigned int, int, int), Rf_allocArray(unsigned int, SEXP);
SEXP Rf_coerceVector(SEXP, unsigned int), Rf_mkChar(const char*), Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_protect(SEXP);
SEXP Rf_ScalarReal(double), Rf_ScalarInteger(int);
void Rprintf(const char*, ...) __attribute__((format(printf, 1, 2)));
void Rf_unprotect(int);
void Rf_error(const char*, ...) __attribute__((noreturn, format(printf, 1, 2)));
int* INTEGER(SEXP);
double* REAL(SEXP);
R_xlen_t XLENGTH(SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t), SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
#define error Rf_error
#define isMatrix Rf_isMatrix
#define nrows Rf_nrows
#define ncols Rf_ncols
#define asInteger Rf_asInteger
#define asLogical Rf_asLogical
#define asReal Rf_asReal
#define allocVector Rf_allocVector
#define allocMatrix Rf_allocMatrix
#define allocArray Rf_allocArray
#define coerceVector Rf_coerceVector
#define mkChar Rf_mkChar
#define setAttrib Rf_setAttrib
#define ScalarReal Rf_ScalarReal
#define ScalarInteger Rf_ScalarInteger
#endif

/* mock_r.c -- TEST-ONLY: a few dozen lines' worth of the R C API, implemented on a tagged struct, so that
 * tests/test_r_shim_exec.py can RUN bmm-mcmc_amd/r-shim/bmmmcmc_shim.c in an image without R: the shim is
 * compiled together with this file and linked against the real libbmmmcmc_hip.so; the test builds SEXPs,
 * calls R_init_bmmmcmc, looks entry points up in the table the shim registered (as .Call does) and
 * inspects the lists that come back.  It implements the API symbols declared in Rinternals.h / R_ext of this
 * directory with the semantics "Writing R Extensions" documents for them -- enough for this one shim; it is
 * not R, is never shipped, and is not used to build or stand in for anything of the reference package.
 *
 * What a test can observe beyond return values: the PROTECT depth (must be back to where it was after every
 * call, error or not), what Rprintf printed, the message of an Rf_error (which longjmps out of the shim, as
 * R's error does), and how many times R's RNG stream was read.
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R_ext/Rdynload.h"
#include "R_ext/Random.h"
#include "Rinternals.h"

#define CHARSXP 9

struct SEXPREC {
    int type;
    R_xlen_t len;
    SEXP dim, names; /* the two attributes the shim touches */
    void* data;      /* int*, double*, SEXP*, or char* (CHARSXP) */
};

static struct SEXPREC nil_rec = {0, 0, NULL, NULL, NULL}, names_sym = {1, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec, R_NamesSymbol = &names_sym;
int R_NaInt = (-2147483647 - 1);

static int g_protect = 0, g_rng_open = 0, g_rng_reads = 0;
static uint64_t g_rng = 12345;
static char g_error[1024], g_printed[1 << 16];
static size_t g_printed_n = 0;
static jmp_buf g_jmp;
static int g_jmp_armed = 0;
static const R_CallMethodDef* g_table = NULL;
static int g_dynamic_symbols = -1;

static size_t elsize(int type) {
    switch (type) {
        case INTSXP: case LGLSXP: return sizeof(int);
        case REALSXP: return sizeof(double);
        case RAWSXP: case CHARSXP: return 1;
        case VECSXP: case STRSXP: return sizeof(SEXP);
    }
    return 0;
}
static SEXP new_vec(int type, R_xlen_t n) {
    SEXP x = (SEXP)calloc(1, sizeof(struct SEXPREC));
    x->type = type; x->len = n; x->dim = R_NilValue; x->names = R_NilValue;
    x->data = calloc((size_t)(n > 0 ? n : 1), elsize(type) ? elsize(type) : 1);
    if (type == VECSXP || type == STRSXP) for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)x->data)[i] = R_NilValue;
    return x;
}

/* ---------------------------------------------------------------- the API the shim uses */
int R_IsNaN(double x) { return x != x; }
int TYPEOF(SEXP x) { return x->type; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
int* INTEGER(SEXP x) { return (int*)x->data; }
double* REAL(SEXP x) { return (double*)x->data; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP*)x->data)[i] = v; return v; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP*)x->data)[i] = v; }
SEXP Rf_allocVector(unsigned int type, R_xlen_t n) { return new_vec((int)type, n); }
SEXP Rf_allocMatrix(unsigned int type, int nr, int nc) {
    SEXP x = new_vec((int)type, (R_xlen_t)nr * nc), d = new_vec(INTSXP, 2);
    INTEGER(d)[0] = nr; INTEGER(d)[1] = nc;
    x->dim = d;
    return x;
}
SEXP Rf_allocArray(unsigned int type, SEXP dims) {
    R_xlen_t n = 1;
    for (R_xlen_t i = 0; i < dims->len; ++i) n *= INTEGER(dims)[i];
    SEXP x = new_vec((int)type, n), d = new_vec(INTSXP, dims->len);
    memcpy(d->data, dims->data, sizeof(int) * (size_t)dims->len);
    x->dim = d;
    return x;
}
SEXP Rf_mkChar(const char* s) {
    SEXP x = new_vec(CHARSXP, (R_xlen_t)strlen(s) + 1);
    strcpy((char*)x->data, s);
    return x;
}
SEXP Rf_ScalarReal(double v) { SEXP x = new_vec(REALSXP, 1); REAL(x)[0] = v; return x; }
SEXP Rf_ScalarInteger(int v) { SEXP x = new_vec(INTSXP, 1); INTEGER(x)[0] = v; return x; }
SEXP Rf_setAttrib(SEXP x, SEXP sym, SEXP v) { if (sym == R_NamesSymbol) x->names = v; return v; }
Rboolean Rf_isMatrix(SEXP x) { return x != R_NilValue && x->dim != R_NilValue && x->dim->len == 2; }
int Rf_nrows(SEXP x) { return Rf_isMatrix(x) ? INTEGER(x->dim)[0] : (int)x->len; }
int Rf_ncols(SEXP x) { return Rf_isMatrix(x) ? INTEGER(x->dim)[1] : 1; }
SEXP Rf_protect(SEXP x) { ++g_protect; return x; }
void Rf_unprotect(int n) { g_protect -= n; }

static double elt_as_real(SEXP x, R_xlen_t i) {
    switch (x->type) {
        case INTSXP: case LGLSXP: return INTEGER(x)[i] == R_NaInt ? (0.0 / 0.0) : (double)INTEGER(x)[i];
        case REALSXP: return REAL(x)[i];
        case RAWSXP: return (double)((unsigned char*)x->data)[i];
    }
    return 0.0 / 0.0;
}
/* atomic -> atomic coercion keeps the attributes, as R's coerceVector does (as.integer() drops them in R
 * code, the C entry point does not); doubles are truncated towards zero, NaN -> NA */
SEXP Rf_coerceVector(SEXP x, unsigned int type) {
    if ((unsigned)x->type == type) return x;
    SEXP y = new_vec((int)type, x->len);
    y->dim = x->dim; y->names = x->names;
    for (R_xlen_t i = 0; i < x->len; ++i) {
        const double v = elt_as_real(x, i);
        if (type == INTSXP || type == LGLSXP) INTEGER(y)[i] = v != v ? R_NaInt : (type == LGLSXP ? v != 0 : (int)v);
        else if (type == REALSXP) REAL(y)[i] = v;
    }
    return y;
}
int Rf_asInteger(SEXP x) {
    if (x == R_NilValue || x->len < 1) return R_NaInt;
    const double v = elt_as_real(x, 0);
    return v != v || v > 2147483647.0 || v < -2147483647.0 ? R_NaInt : (int)v;
}
int Rf_asLogical(SEXP x) {
    if (x == R_NilValue || x->len < 1) return R_NaInt;
    const double v = elt_as_real(x, 0);
    return v != v ? R_NaInt : v != 0;
}
double Rf_asReal(SEXP x) { return x == R_NilValue || x->len < 1 ? (0.0 / 0.0) : elt_as_real(x, 0); }

void Rf_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    if (!g_jmp_armed) { fprintf(stderr, "mock R: error outside a call: %s\n", g_error); abort(); }
    longjmp(g_jmp, 1);
}
void Rprintf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (g_printed_n < sizeof g_printed - 1) {
        int n = vsnprintf(g_printed + g_printed_n, sizeof g_printed - g_printed_n, fmt, ap);
        if (n > 0) g_printed_n += (size_t)n < sizeof g_printed - g_printed_n ? (size_t)n : sizeof g_printed - g_printed_n - 1;
    }
    va_end(ap);
}
void GetRNGstate(void) { ++g_rng_open; }
void PutRNGstate(void) { --g_rng_open; }
double unif_rand(void) { /* reads are only legal between Get and Put; a 53-bit LCG stands in for the stream */
    if (g_rng_open != 1) { snprintf(g_error, sizeof g_error, "unif_rand outside GetRNGstate/PutRNGstate"); abort(); }
    ++g_rng_reads;
    g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(g_rng >> 11) * (1.0 / 9007199254740992.0);
}
int R_registerRoutines(DllInfo* dll, const void* c, const R_CallMethodDef* call, const void* f, const void* e) {
    g_table = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* dll, Rboolean v) { g_dynamic_symbols = v; return TRUE; }

/* ---------------------------------------------------------------- what the Python test drives */
extern void R_init_bmmmcmc(DllInfo* dll);
void mock_init(void) { g_table = NULL; R_init_bmmmcmc(NULL); }
int mock_dynamic_symbols(void) { return g_dynamic_symbols; }
int mock_registered(int i, const char** name, int* arity) { /* row i of the registered table; 0 past the end */
    if (!g_table) return 0;
    for (int k = 0; k <= i; ++k) if (!g_table[k].name) return 0;
    *name = g_table[i].name; *arity = g_table[i].numArgs;
    return 1;
}
SEXP mock_nil(void) { return R_NilValue; }
SEXP mock_int(const int* v, R_xlen_t n, int nr, int nc) { /* nr > 0: a matrix */
    SEXP x = nr > 0 ? Rf_allocMatrix(INTSXP, nr, nc) : new_vec(INTSXP, n);
    memcpy(x->data, v, sizeof(int) * (size_t)n);
    return x;
}
SEXP mock_real(const double* v, R_xlen_t n, int nr, int nc) {
    SEXP x = nr > 0 ? Rf_allocMatrix(REALSXP, nr, nc) : new_vec(REALSXP, n);
    memcpy(x->data, v, sizeof(double) * (size_t)n);
    return x;
}
void mock_set_type(SEXP x, int type) { x->type = type; } /* INTSXP <-> LGLSXP share storage */
SEXP mock_lgl(int v) { SEXP x = new_vec(LGLSXP, 1); INTEGER(x)[0] = v; return x; }
SEXP mock_str(const char* s) { SEXP x = new_vec(STRSXP, 1); SET_STRING_ELT(x, 0, Rf_mkChar(s)); return x; }
int mock_type(SEXP x) { return x->type; }
long long mock_length(SEXP x) { return (long long)x->len; }
void* mock_data(SEXP x) { return x->data; }
int mock_ndim(SEXP x) { return x->dim == R_NilValue ? 0 : (int)x->dim->len; }
int mock_dim(SEXP x, int k) { return INTEGER(x->dim)[k]; }
const char* mock_name(SEXP x, int i) { return x->names == R_NilValue ? NULL : (const char*)VECTOR_ELT(x->names, i)->data; }
SEXP mock_elt(SEXP x, int i) { return VECTOR_ELT(x, i); }
const char* mock_last_error(void) { return g_error; }
const char* mock_printed(void) { g_printed[g_printed_n] = 0; return g_printed; }
void mock_clear_printed(void) { g_printed_n = 0; }
int mock_protect_depth(void) { return g_protect; }
int mock_rng_reads(void) { return g_rng_reads; }
void mock_set_seed(unsigned long long s) { g_rng = s; }

typedef SEXP (*fn1)(SEXP);
/* .Call(name, args...): the registered routine of that name, with the arity it was registered with; NULL
 * (and mock_last_error) when the name is unknown, the arity differs, or the routine raised an R error */
SEXP mock_call(const char* name, int nargs, SEXP* a) {
    g_error[0] = 0;
    const R_CallMethodDef* e = g_table;
    while (e && e->name && strcmp(e->name, name) != 0) ++e;
    if (!e || !e->name) { snprintf(g_error, sizeof g_error, "\"%s\" not available for .Call() for package \"bmmmcmc\"", name); return NULL; }
    if (e->numArgs != nargs) { snprintf(g_error, sizeof g_error, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, e->numArgs, name); return NULL; }
    const int depth = g_protect;
    SEXP r = NULL;
    g_jmp_armed = 1;
    if (setjmp(g_jmp) == 0) {
        DL_FUNC f = e->fun;
#define A(i) a[i]
        switch (nargs) {
            case 1: r = ((SEXP(*)(SEXP))f)(A(0)); break;
            case 2: r = ((SEXP(*)(SEXP, SEXP))f)(A(0), A(1)); break;
            case 12: r = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(
                         A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11)); break;
            case 13: r = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(
                         A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12)); break;
            case 14: r = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(
                         A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13)); break;
            case 16: r = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(
                         A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15)); break;
            case 17: r = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(
                         A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16)); break;
            default: snprintf(g_error, sizeof g_error, "mock R: no trampoline for %d arguments", nargs); r = NULL;
        }
#undef A
    } else {
        r = NULL;          /* an R error: R unwinds the protect stack to where the call started */
        g_protect = depth;
        g_rng_open = 0;
    }
    g_jmp_armed = 0;
    return r;
}

/* _lsr_pyhelper: host-side glue of the Python binding (liblsr_hip.so itself has no Python dependency).
 *   - the data pointers and lengths of a Python list of NumPy arrays in one pass (arr.ctypes.data
 *     costs about 2 us per array in Python; a batch has 10 000 clips);
 *   - the factorisation of a batch of covariance matrices on a pool of threads, calling the LAPACK
 *     routines SciPy and NumPy call (through the function pointers SciPy exports) so that the results
 *     carry the same bits as scipy.linalg.eigh / numpy.linalg.qr, which hold the GIL or run one
 *     matrix after the other. */
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <numpy/arrayobject.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* clip_table(seq, ptrs, lens) -> int
 * seq: sequence of arrays; ptrs: writable uint64[n] buffer; lens: writable int64[n] buffer.
 * Returns NPY_INT16's or NPY_FLOAT32's type number when every element is a 1-D C-contiguous aligned
 * array of that one type (tables filled), else -1 (tables undefined: the caller converts and retries). */
static PyObject* clip_table(PyObject* self, PyObject* args) {
    PyObject* seq;
    Py_buffer ptrs, lens;
    if (!PyArg_ParseTuple(args, "Ow*w*", &seq, &ptrs, &lens)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "clip_table: expected a sequence of arrays");
    long result = -1;
    if (fast) {
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
        if (ptrs.len < (Py_ssize_t)(n * sizeof(uint64_t)) || lens.len < (Py_ssize_t)(n * sizeof(int64_t))) {
            PyErr_SetString(PyExc_ValueError, "clip_table: output buffers too small");
            Py_DECREF(fast);
            PyBuffer_Release(&ptrs);
            PyBuffer_Release(&lens);
            return NULL;
        }
        uint64_t* p = (uint64_t*)ptrs.buf;
        int64_t* l = (int64_t*)lens.buf;
        int type = -2;
        PyObject** items = PySequence_Fast_ITEMS(fast);
        Py_ssize_t i = 0;
        for (; i < n; ++i) {
            if (!PyArray_Check(items[i])) break;
            PyArrayObject* a = (PyArrayObject*)items[i];
            const int t = PyArray_TYPE(a);
            if (type == -2) type = t;
            if (t != type || (t != NPY_INT16 && t != NPY_FLOAT32) || PyArray_NDIM(a) != 1 ||
                !PyArray_IS_C_CONTIGUOUS(a) || !PyArray_ISALIGNED(a) || !PyArray_ISNOTSWAPPED(a))
                break;
            p[i] = (uint64_t)(uintptr_t)PyArray_DATA(a);
            l[i] = (int64_t)PyArray_DIM(a, 0);
        }
        if (i == n && n > 0) result = type;
        Py_DECREF(fast);
    }
    PyBuffer_Release(&ptrs);
    PyBuffer_Release(&lens);
    if (!fast) return NULL;
    return PyLong_FromLong(result);
}

/* array_table(seq, ptrs, sizes, ndim, inner) -> int
 * Like clip_table for C-contiguous arrays of `ndim` dimensions whose last dimension is `inner` (ignored when
 * ndim == 1): sizes[i] = number of ELEMENTS of array i.  Same return convention. */
static PyObject* array_table(PyObject* self, PyObject* args) {
    PyObject* seq;
    Py_buffer ptrs, lens;
    int ndim, inner;
    if (!PyArg_ParseTuple(args, "Ow*w*ii", &seq, &ptrs, &lens, &ndim, &inner)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "array_table: expected a sequence of arrays");
    long result = -1;
    if (fast) {
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
        if (ptrs.len < (Py_ssize_t)(n * sizeof(uint64_t)) || lens.len < (Py_ssize_t)(n * sizeof(int64_t))) {
            PyErr_SetString(PyExc_ValueError, "array_table: output buffers too small");
            Py_DECREF(fast);
            PyBuffer_Release(&ptrs);
            PyBuffer_Release(&lens);
            return NULL;
        }
        uint64_t* p = (uint64_t*)ptrs.buf;
        int64_t* l = (int64_t*)lens.buf;
        int type = -2;
        PyObject** items = PySequence_Fast_ITEMS(fast);
        Py_ssize_t i = 0;
        for (; i < n; ++i) {
            if (!PyArray_Check(items[i])) break;
            PyArrayObject* a = (PyArrayObject*)items[i];
            const int t = PyArray_TYPE(a);
            if (type == -2) type = t;
            if (t != type || (t != NPY_INT16 && t != NPY_FLOAT32) || PyArray_NDIM(a) != ndim ||
                (ndim > 1 && PyArray_DIM(a, ndim - 1) != inner) || !PyArray_IS_C_CONTIGUOUS(a) || !PyArray_ISALIGNED(a) ||
                !PyArray_ISNOTSWAPPED(a))
                break;
            p[i] = (uint64_t)(uintptr_t)PyArray_DATA(a);
            l[i] = (int64_t)PyArray_SIZE(a);
        }
        if (i == n && n > 0) result = type;
        Py_DECREF(fast);
    }
    PyBuffer_Release(&ptrs);
    PyBuffer_Release(&lens);
    if (!fast) return NULL;
    return PyLong_FromLong(result);
}

/* ---- batched covariance factorisation ------------------------------------------------------------------
 * For every selected matrix C (symmetric, row-major, only the lower triangle is read, as
 * scipy.linalg.eigh(C, lower=True) does):
 *   lam, V   = dsyevr('V', 'A', 'L') with the workspace sizes of its own query  (scipy.linalg.eigh, driver "evr")
 *   U[k][j]  = V[k][j] * sqrt(1 / lam[j])                                       (the whitening matrix SciPy's frozen
 *                                                                                multivariate_normal keeps)
 *   L        = R^T of U^T = Q R (dgeqrf on U^T, upper triangle kept)            (numpy.linalg.qr(U.T, mode="r").T)
 * status = LAPACK's info of dsyevr (or -1000 when it found fewer than n eigenvalues, -2000 - info for dgeqrf).
 * Matrices first, first + step, ... are processed; the rows of the others are left untouched. */
/* LAPACK's integer arguments are 32 bits wide in SciPy's library (LP64) and 64 bits in the ILP64 build NumPy ships:
 * the routines are held as untyped pointers and called through the signature that matches (lapack_int). */
typedef void (*lapack_fn)(void);

struct factor_lib {
    lapack_fn syevr, geqrf;
    int ilp64;
    int lwork, liwork, qwork;     /* workspace sizes of this library's own queries */
};

struct factor_job {
    struct factor_lib lib[2];      /* lib[1].syevr == NULL: one library only */
    const double* cov;
    double *lam, *whiten, *tri;
    int32_t* status;
    int n, n_mats, first, step;
    int next;            /* next share to hand out (atomic) */
    int failed;          /* allocation failure */
};

#define FACTOR_DEFINE(SUFFIX, INT)                                                                                  \
    typedef void (*dsyevr_##SUFFIX)(char*, char*, char*, INT*, double*, INT*, double*, double*, INT*, INT*, double*,  \
                                    INT*, double*, double*, INT*, INT*, double*, INT*, INT*, INT*, INT*);            \
    typedef void (*dgeqrf_##SUFFIX)(INT*, INT*, double*, INT*, double*, double*, INT*, INT*);                        \
    /* workspace sizes, as scipy.linalg.eigh (dsyevr_lwork) and numpy.linalg.qr (lwork = -1) ask for them */         \
    static int factor_query_##SUFFIX(struct factor_lib* L, int n_) {                                                 \
        char jobz = 'V', range = 'A', uplo = 'L';                                                                    \
        INT n = n_, lda = n_, ldz = n_, il = 1, iu = n_, m = 0, info = 0, lwork = -1, liwork = -1, iwq = 0, isup = 0; \
        double vl = 0.0, vu = 1.0, abstol = 0.0, wq = 0.0, dummy = 0.0;                                              \
        ((dsyevr_##SUFFIX)L->syevr)(&jobz, &range, &uplo, &n, &dummy, &lda, &vl, &vu, &il, &iu, &abstol, &m, &dummy,  \
                                    &dummy, &ldz, &isup, &wq, &lwork, &iwq, &liwork, &info);                         \
        if (info != 0) return (int)info;                                                                             \
        INT q_lwork = -1, qinfo = 0;                                                                                 \
        double qq = 0.0;                                                                                             \
        ((dgeqrf_##SUFFIX)L->geqrf)(&n, &n, &dummy, &lda, &dummy, &qq, &q_lwork, &qinfo);                             \
        if (qinfo != 0) return (int)qinfo;                                                                           \
        L->lwork = (int)wq > 1 ? (int)wq : 1;                                                                        \
        L->liwork = (int)iwq > 1 ? (int)iwq : 1;                                                                     \
        L->qwork = (int)qq > n_ ? (int)qq : n_;                                                                      \
        return 0;                                                                                                    \
    }                                                                                                                \
    /* one matrix with the scratch memory at a (doubles) / iw (INTs) */                                              \
    static void factor_one_##SUFFIX(const struct factor_job* J, const struct factor_lib* L, int i, double* a,        \
                                    void* iw_) {                                                                     \
        const int n = J->n;                                                                                          \
        const size_t nn = (size_t)n * n;                                                                             \
        double *z = a + nn, *work = z + nn, *qw = work + L->lwork, *w = qw + L->qwork, *tau = w + n;                  \
        INT *iwork = (INT*)iw_, *isuppz = iwork + L->liwork;                                                         \
        const double* c = J->cov + (size_t)i * nn;                                                                   \
        /* column-major copy of the row-major matrix: a[r + col * n] = C[r][col] */                                  \
        for (int r = 0; r < n; ++r)                                                                                  \
            for (int col = 0; col < n; ++col) a[r + (size_t)col * n] = c[(size_t)r * n + col];                       \
        char jobz = 'V', range = 'A', uplo = 'L';                                                                    \
        INT nn_ = n, lda = n, ldz = n, il = 1, iu = n, m = 0, info = 0, lwork = L->lwork, liwork = L->liwork;         \
        double vl = 0.0, vu = 1.0, abstol = 0.0;                                                                     \
        ((dsyevr_##SUFFIX)L->syevr)(&jobz, &range, &uplo, &nn_, a, &lda, &vl, &vu, &il, &iu, &abstol, &m, w, z, &ldz, \
                                    isuppz, work, &lwork, iwork, &liwork, &info);                                    \
        int status = (int)info;                                                                                      \
        if (status == 0 && m != n) status = -1000;                                                                   \
        double* U = J->whiten + (size_t)i * nn;                                                                      \
        double* Lt = J->tri + (size_t)i * nn;                                                                        \
        for (int j = 0; j < n; ++j) {                                                                                \
            J->lam[(size_t)i * n + j] = w[j];                                                                        \
            const double scale = sqrt(1.0 / w[j]);                                                                   \
            for (int k = 0; k < n; ++k) U[(size_t)k * n + j] = z[k + (size_t)j * n] * scale;                         \
        }                                                                                                            \
        if (status == 0) {                                                                                           \
            /* the row-major U read as column-major is U^T: factor it in place of a copy */                          \
            memcpy(Lt, U, sizeof(double) * nn);                                                                      \
            INT qinfo = 0, qwork = L->qwork;                                                                         \
            ((dgeqrf_##SUFFIX)L->geqrf)(&nn_, &nn_, Lt, &lda, tau, qw, &qwork, &qinfo);                               \
            if (qinfo != 0) status = -2000 - (int)qinfo;                                                             \
            /* R[j][k] (j <= k) sits at Lt[j + k * n] = row-major L[k][j]: zero what lies above the diagonal of L */ \
            for (int k = 0; k < n; ++k)                                                                              \
                for (int j = k + 1; j < n; ++j) Lt[(size_t)k * n + j] = 0.0;                                         \
        } else {                                                                                                     \
            memset(Lt, 0, sizeof(double) * nn);                                                                      \
        }                                                                                                            \
        J->status[i] = status;                                                                                       \
    }

FACTOR_DEFINE(lp64, int)
FACTOR_DEFINE(ilp64, int64_t)

/* takes matrices off the job until none are left; scratch memory is the caller's and grows on demand */
struct factor_scratch {
    double* a;
    int* iw;
    size_t a_len, iw_len;
};

static void factor_drain(struct factor_job* J, struct factor_scratch* S, int who) {
    /* with two libraries the workers alternate between them: each library serialises part of every BLAS call on a
     * lock of its own, and two libraries are two locks */
    const struct factor_lib* L = &J->lib[J->lib[1].syevr ? (who & 1) : 0];
    const size_t nn = (size_t)J->n * J->n;
    const size_t need_a = 2 * nn + (size_t)L->lwork + (size_t)L->qwork + 2 * (size_t)J->n;
    const size_t need_i = 2 * ((size_t)L->liwork + 2 * (size_t)J->n);      /* in ints: room for 64-bit integers */
    if (S->a_len < need_a) {
        free(S->a);
        S->a = (double*)malloc(sizeof(double) * need_a);
        S->a_len = S->a ? need_a : 0;
    }
    if (S->iw_len < need_i) {
        free(S->iw);
        S->iw = (int*)malloc(sizeof(int) * need_i);
        S->iw_len = S->iw ? need_i : 0;
    }
    if (!S->a || !S->iw) {
        __atomic_store_n(&J->failed, 1, __ATOMIC_RELAXED);
        return;
    }
    for (;;) {
        const int share = __atomic_fetch_add(&J->next, 1, __ATOMIC_RELAXED);
        const long i = (long)J->first + (long)J->step * share;
        if (i >= J->n_mats) break;
        if (L->ilp64) factor_one_ilp64(J, L, (int)i, S->a, S->iw);
        else factor_one_lp64(J, L, (int)i, S->a, S->iw);
    }
}

/* Worker threads are started once and wait for jobs: starting sixteen threads costs about as much as the
 * factorisations of a 58-state model take.  A child process after fork() has none of them and starts its own. */
#define FACTOR_MAX_THREADS 64
static struct {
    pthread_mutex_t mu;
    pthread_cond_t wake, idle;
    pthread_t tid[FACTOR_MAX_THREADS];
    int n_workers;           /* threads started */
    int wanted;              /* workers that should join the current job */
    int running;             /* workers still inside the current job */
    unsigned generation;
    struct factor_job* job;
    pid_t pid;
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, NULL, 0};

static void* factor_worker(void* arg) {
    const int me = (int)(intptr_t)arg;
    struct factor_scratch S = {NULL, NULL, 0, 0};
    unsigned seen = 0;
    pthread_mutex_lock(&pool.mu);
    for (;;) {
        while (pool.generation == seen) pthread_cond_wait(&pool.wake, &pool.mu);
        seen = pool.generation;
        if (me >= pool.wanted) continue;
        struct factor_job* J = pool.job;
        pthread_mutex_unlock(&pool.mu);
        factor_drain(J, &S, me + 1);
        pthread_mutex_lock(&pool.mu);
        if (--pool.running == 0) pthread_cond_signal(&pool.idle);
    }
    return NULL;
}

static pthread_mutex_t call_mu = PTHREAD_MUTEX_INITIALIZER;      /* one factor_batch call at a time */

/* runs the job on the calling thread plus `helpers` pool workers (fewer if threads cannot be started) */
static void factor_run(struct factor_job* J, int helpers) {
    static struct factor_scratch mine = {NULL, NULL, 0, 0};       /* callers hold call_mu: one at a time */
    if (helpers > FACTOR_MAX_THREADS) helpers = FACTOR_MAX_THREADS;
    pthread_mutex_lock(&pool.mu);
    if (pool.pid != getpid()) {              /* first use, or a forked child: no workers exist in this process */
        pool.pid = getpid();
        pool.n_workers = 0;
        pool.running = 0;
    }
    while (pool.n_workers < helpers) {
        if (pthread_create(&pool.tid[pool.n_workers], NULL, factor_worker, (void*)(intptr_t)pool.n_workers) != 0) break;
        pthread_detach(pool.tid[pool.n_workers]);
        ++pool.n_workers;
    }
    if (helpers > pool.n_workers) helpers = pool.n_workers;
    pool.job = J;
    pool.wanted = helpers;
    pool.running = helpers;
    ++pool.generation;
    if (helpers > 0) pthread_cond_broadcast(&pool.wake);
    pthread_mutex_unlock(&pool.mu);
    factor_drain(J, &mine, 0);
    pthread_mutex_lock(&pool.mu);
    while (pool.running > 0) pthread_cond_wait(&pool.idle, &pool.mu);
    pool.job = NULL;
    pthread_mutex_unlock(&pool.mu);
}

/* a LAPACK routine handed over as the PyCapsule scipy.linalg.cython_lapack exports, or as an integer address */
static lapack_fn routine_pointer(PyObject* obj) {
    if (PyCapsule_CheckExact(obj)) return (lapack_fn)PyCapsule_GetPointer(obj, PyCapsule_GetName(obj));
    if (PyLong_Check(obj)) {
        void* p = PyLong_AsVoidPtr(obj);
        if (!p && !PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "factor_batch: null routine address");
        return (lapack_fn)p;
    }
    PyErr_SetString(PyExc_TypeError, "factor_batch: expected the PyCapsule or the address of a LAPACK routine");
    return NULL;
}

/* factor_batch(dsyevr, dgeqrf, cov, lam, whiten, tri, status, n_mats, n, first, step, threads[, dsyevr2, dgeqrf2, ilp64_2])
 * -> None.  The optional second pair names the same two routines in a second LAPACK library (ilp64_2: its integer
 * arguments are 64 bits wide); the worker threads then alternate between the two libraries. */
static PyObject* factor_batch(PyObject* self, PyObject* args) {
    PyObject *o_syevr, *o_geqrf, *o_syevr2 = NULL, *o_geqrf2 = NULL;
    Py_buffer cov, lam, whiten, tri, status;
    int n_mats, n, first, step, threads, ilp64_2 = 0;
    if (!PyArg_ParseTuple(args, "OOy*w*w*w*w*iiiii|OOi", &o_syevr, &o_geqrf, &cov, &lam, &whiten, &tri, &status, &n_mats, &n,
                          &first, &step, &threads, &o_syevr2, &o_geqrf2, &ilp64_2))
        return NULL;
    PyObject* result = NULL;
    struct factor_job J;
    memset(&J, 0, sizeof(J));
    J.lib[0].syevr = routine_pointer(o_syevr);
    J.lib[0].geqrf = J.lib[0].syevr ? routine_pointer(o_geqrf) : NULL;
    if (!J.lib[0].syevr || !J.lib[0].geqrf) goto done;
    if (o_syevr2 && o_syevr2 != Py_None) {
        J.lib[1].syevr = routine_pointer(o_syevr2);
        J.lib[1].geqrf = J.lib[1].syevr && o_geqrf2 ? routine_pointer(o_geqrf2) : NULL;
        J.lib[1].ilp64 = ilp64_2 != 0;
        if (!J.lib[1].syevr || !J.lib[1].geqrf) goto done;
    }
    {
        const Py_ssize_t nn = (Py_ssize_t)n * n;
        if (n < 1 || n_mats < 0 || first < 0 || step < 1 || cov.len < (Py_ssize_t)(n_mats * nn * sizeof(double)) ||
            whiten.len < cov.len || tri.len < cov.len || lam.len < (Py_ssize_t)(n_mats * (Py_ssize_t)n * sizeof(double)) ||
            status.len < (Py_ssize_t)(n_mats * sizeof(int32_t))) {
            PyErr_SetString(PyExc_ValueError, "factor_batch: buffer sizes do not match n_mats x n x n");
            goto done;
        }
        for (int k = 0; k < 2 && J.lib[k].syevr; ++k) {
            const int info = J.lib[k].ilp64 ? factor_query_ilp64(&J.lib[k], n) : factor_query_lp64(&J.lib[k], n);
            if (info != 0) {
                PyErr_Format(PyExc_RuntimeError, "factor_batch: workspace query of library %d failed (info %d)", k, info);
                goto done;
            }
        }
        int count = (n_mats - first + step - 1) / step;
        if (count < 0) count = 0;
        if (threads < 1) threads = 1;
        if (threads > count) threads = count > 0 ? count : 1;
        J.cov = (const double*)cov.buf;
        J.lam = (double*)lam.buf;
        J.whiten = (double*)whiten.buf;
        J.tri = (double*)tri.buf;
        J.status = (int32_t*)status.buf;
        J.n = n;
        J.n_mats = n_mats;
        J.first = first;
        J.step = step;
        /* the pool and the caller's scratch are process-wide: one call at a time; the GIL is released while the
         * calling thread waits for its turn and works on its share */
        Py_BEGIN_ALLOW_THREADS
        pthread_mutex_lock(&call_mu);
        factor_run(&J, threads - 1);
        pthread_mutex_unlock(&call_mu);
        Py_END_ALLOW_THREADS
        if (J.failed) {
            PyErr_NoMemory();
            goto done;
        }
    }
    result = Py_None;
    Py_INCREF(result);
done:
    PyBuffer_Release(&cov);
    PyBuffer_Release(&lam);
    PyBuffer_Release(&whiten);
    PyBuffer_Release(&tri);
    PyBuffer_Release(&status);
    return result;
}

static PyMethodDef methods[] = {
    {"factor_batch", factor_batch, METH_VARARGS, "eigendecomposition, whitening and triangular factor of a batch of covariances"},
    {"array_table", array_table, METH_VARARGS, "fill pointer and size tables of a list of n-D int16/float32 arrays"},
    {"clip_table", clip_table, METH_VARARGS, "fill pointer and length tables of a list of 1-D int16/float32 arrays"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_lsr_pyhelper", NULL, -1, methods};

PyMODINIT_FUNC PyInit__lsr_pyhelper(void) {
    import_array();
    PyObject* m = PyModule_Create(&module);
    if (m) {
        PyModule_AddIntConstant(m, "INT16", NPY_INT16);
        PyModule_AddIntConstant(m, "FLOAT32", NPY_FLOAT32);
    }
    return m;
}

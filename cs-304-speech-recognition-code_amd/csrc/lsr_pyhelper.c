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
typedef void (*dsyevr_fn)(char*, char*, char*, int*, double*, int*, double*, double*, int*, int*, double*, int*, double*,
                          double*, int*, int*, double*, int*, int*, int*, int*);
typedef void (*dgeqrf_fn)(int*, int*, double*, int*, double*, double*, int*, int*);

struct factor_job {
    dsyevr_fn syevr;
    dgeqrf_fn geqrf;
    const double* cov;
    double *lam, *whiten, *tri;
    int32_t* status;
    int n, n_mats, first, step;
    int lwork, liwork, qwork;
    int next;            /* next share to hand out (atomic) */
    int failed;          /* allocation failure */
};

/* one matrix with the scratch memory at a (doubles) / iw (ints) */
static void factor_one(const struct factor_job* J, int i, double* a, int* iw) {
    const int n = J->n;
    const size_t nn = (size_t)n * n;
    double *z = a + nn, *work = z + nn, *qw = work + J->lwork, *w = qw + J->qwork, *tau = w + n;
    int *iwork = iw, *isuppz = iw + J->liwork;
    const double* c = J->cov + (size_t)i * nn;
    /* column-major copy of the row-major matrix: a[r + col * n] = C[r][col] */
    for (int r = 0; r < n; ++r)
        for (int col = 0; col < n; ++col) a[r + (size_t)col * n] = c[(size_t)r * n + col];
    char jobz = 'V', range = 'A', uplo = 'L';
    int nn_ = n, lda = n, ldz = n, il = 1, iu = n, m = 0, info = 0, lwork = J->lwork, liwork = J->liwork;
    double vl = 0.0, vu = 1.0, abstol = 0.0;
    J->syevr(&jobz, &range, &uplo, &nn_, a, &lda, &vl, &vu, &il, &iu, &abstol, &m, w, z, &ldz, isuppz, work, &lwork, iwork,
             &liwork, &info);
    if (info == 0 && m != n) info = -1000;
    double* U = J->whiten + (size_t)i * nn;
    double* L = J->tri + (size_t)i * nn;
    for (int j = 0; j < n; ++j) {
        J->lam[(size_t)i * n + j] = w[j];
        const double scale = sqrt(1.0 / w[j]);
        for (int k = 0; k < n; ++k) U[(size_t)k * n + j] = z[k + (size_t)j * n] * scale;
    }
    if (info == 0) {
        /* the row-major U read as column-major is U^T: factor it in place of a copy */
        memcpy(L, U, sizeof(double) * nn);
        int qinfo = 0, qwork = J->qwork;
        J->geqrf(&nn_, &nn_, L, &lda, tau, qw, &qwork, &qinfo);
        if (qinfo != 0) info = -2000 - qinfo;
        /* R[j][k] (j <= k) sits at L[j + k * n] = row-major L[k][j]: zero what lies above the diagonal of L */
        for (int k = 0; k < n; ++k)
            for (int j = k + 1; j < n; ++j) L[(size_t)k * n + j] = 0.0;
    } else {
        memset(L, 0, sizeof(double) * nn);
    }
    J->status[i] = info;
}

/* takes matrices off the job until none are left; scratch memory is the caller's and grows on demand */
struct factor_scratch {
    double* a;
    int* iw;
    size_t a_len, iw_len;
};

static void factor_drain(struct factor_job* J, struct factor_scratch* S) {
    const size_t nn = (size_t)J->n * J->n;
    const size_t need_a = 2 * nn + (size_t)J->lwork + (size_t)J->qwork + 2 * (size_t)J->n;
    const size_t need_i = (size_t)J->liwork + 2 * (size_t)J->n;
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
        factor_one(J, (int)i, S->a, S->iw);
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
        factor_drain(J, &S);
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
    factor_drain(J, &mine);
    pthread_mutex_lock(&pool.mu);
    while (pool.running > 0) pthread_cond_wait(&pool.idle, &pool.mu);
    pool.job = NULL;
    pthread_mutex_unlock(&pool.mu);
}

static void* capsule_pointer(PyObject* cap) {
    if (!PyCapsule_CheckExact(cap)) {
        PyErr_SetString(PyExc_TypeError, "factor_batch: expected the PyCapsule of a LAPACK routine");
        return NULL;
    }
    return PyCapsule_GetPointer(cap, PyCapsule_GetName(cap));
}

/* factor_batch(dsyevr, dgeqrf, cov, lam, whiten, tri, status, n_mats, n, first, step, threads) -> None */
static PyObject* factor_batch(PyObject* self, PyObject* args) {
    PyObject *cap_syevr, *cap_geqrf;
    Py_buffer cov, lam, whiten, tri, status;
    int n_mats, n, first, step, threads;
    if (!PyArg_ParseTuple(args, "OOy*w*w*w*w*iiiii", &cap_syevr, &cap_geqrf, &cov, &lam, &whiten, &tri, &status, &n_mats, &n,
                          &first, &step, &threads))
        return NULL;
    PyObject* result = NULL;
    dsyevr_fn syevr = (dsyevr_fn)capsule_pointer(cap_syevr);
    dgeqrf_fn geqrf = syevr ? (dgeqrf_fn)capsule_pointer(cap_geqrf) : NULL;
    if (!syevr || !geqrf) goto done;
    const Py_ssize_t nn = (Py_ssize_t)n * n;
    if (n < 1 || n_mats < 0 || first < 0 || step < 1 || cov.len < (Py_ssize_t)(n_mats * nn * sizeof(double)) ||
        whiten.len < cov.len || tri.len < cov.len || lam.len < (Py_ssize_t)(n_mats * (Py_ssize_t)n * sizeof(double)) ||
        status.len < (Py_ssize_t)(n_mats * sizeof(int32_t))) {
        PyErr_SetString(PyExc_ValueError, "factor_batch: buffer sizes do not match n_mats x n x n");
        goto done;
    }
    {
        /* workspace queries, as scipy.linalg.eigh (dsyevr_lwork) and numpy.linalg.qr (lwork = -1) make them */
        char jobz = 'V', range = 'A', uplo = 'L';
        int nn_ = n, lda = n, ldz = n, il = 1, iu = n, m = 0, info = 0, lwork = -1, liwork = -1, iwq = 0, isup = 0;
        double vl = 0.0, vu = 1.0, abstol = 0.0, wq = 0.0, dummy = 0.0;
        syevr(&jobz, &range, &uplo, &nn_, &dummy, &lda, &vl, &vu, &il, &iu, &abstol, &m, &dummy, &dummy, &ldz, &isup, &wq,
              &lwork, &iwq, &liwork, &info);
        if (info != 0) {
            PyErr_Format(PyExc_RuntimeError, "factor_batch: dsyevr workspace query failed (info %d)", info);
            goto done;
        }
        int q_lwork = -1, qinfo = 0;
        double qq = 0.0;
        geqrf(&nn_, &nn_, &dummy, &lda, &dummy, &qq, &q_lwork, &qinfo);
        if (qinfo != 0) {
            PyErr_Format(PyExc_RuntimeError, "factor_batch: dgeqrf workspace query failed (info %d)", qinfo);
            goto done;
        }
        int count = (n_mats - first + step - 1) / step;
        if (count < 0) count = 0;
        if (threads < 1) threads = 1;
        if (threads > count) threads = count > 0 ? count : 1;
        struct factor_job J;
        memset(&J, 0, sizeof(J));
        J.syevr = syevr;
        J.geqrf = geqrf;
        J.cov = (const double*)cov.buf;
        J.lam = (double*)lam.buf;
        J.whiten = (double*)whiten.buf;
        J.tri = (double*)tri.buf;
        J.status = (int32_t*)status.buf;
        J.n = n;
        J.n_mats = n_mats;
        J.first = first;
        J.step = step;
        J.lwork = (int)wq > 1 ? (int)wq : 1;
        J.liwork = iwq > 1 ? iwq : 1;
        J.qwork = (int)qq > n ? (int)qq : n;
        /* the pool and the caller's scratch are process-wide: the GIL is kept until the job has been handed over
         * and is released while the calling thread works on its share (factor_run) */
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

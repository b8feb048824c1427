/* _lsr_pyhelper: the data pointers and lengths of a Python list of 1-D NumPy arrays in one pass
 * (arr.ctypes.data costs about 2 us per array in Python; a batch has 10 000 clips).  Host-side
 * glue of the Python binding only: liblsr_hip.so itself has no Python dependency. */
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <numpy/arrayobject.h>
#include <stdint.h>

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

static PyMethodDef methods[] = {
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

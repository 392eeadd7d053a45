/* psk_pylist.c -- host-side packer for python LISTS of keys (CPython extension `pyprobables_amd._pylist`).
 *
 * The reference hashes one python object per call: fnv_1a walks `list(key)` for bytes-likes and `map(ord, key)` for str
 * (probables/hashes.py:98), i.e. a key is its sequence of byte values / code points.  add_many(list_of_keys) has to turn a
 * million small objects into ONE buffer before the engine sees them; in python that is "".join + map(len) (keys.py
 * _pack_homogeneous: 25-65 ns per key, what bounds add_many(list)).  Here: two walks over the list with the C API --
 * lengths and the widest element first, then one memcpy per key straight out of the objects' own storage (a compact str of
 * code points <= 255 IS its latin-1 bytes: PyUnicode_1BYTE_DATA).
 *
 *   pack(list) -> None                                  (an element that is not str / bytes / bytearray: the python path decides / raises)
 *              -> (layout, blob: bytearray, offsets: bytearray | None, n, key_len)
 *                   layout 0 = PSK_KEYS_FIXED    uint8[n][key_len]              (all keys of one length, no code point > 255)
 *                          1 = PSK_KEYS_VARLEN8  uint8 blob + uint64 offsets[n + 1]
 *                          2 = PSK_KEYS_VARLEN32 uint32 code points + uint64 offsets[n + 1]   (some code point > 255: every key widened)
 *
 * Host plumbing, not the product path: no hashing happens here, and keys.py falls back to its python packer when the module is absent.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static int key_view(PyObject *o, const void **data, Py_ssize_t *len, int *kind)
{
    if (PyUnicode_Check(o)) {
        if (PyUnicode_READY(o) < 0) return -1;
        *kind = PyUnicode_KIND(o);  /* 1, 2 or 4 bytes per code point */
        *data = PyUnicode_DATA(o);
        *len = PyUnicode_GET_LENGTH(o);
        return 0;
    }
    if (PyBytes_Check(o)) {
        *kind = 1;
        *data = PyBytes_AS_STRING(o);
        *len = PyBytes_GET_SIZE(o);
        return 0;
    }
    if (PyByteArray_Check(o)) {
        *kind = 1;
        *data = PyByteArray_AS_STRING(o);
        *len = PyByteArray_GET_SIZE(o);
        return 0;
    }
    return 1; /* not ours */
}

static PyObject *pack(PyObject *self, PyObject *arg)
{
    (void)self;
    if (!PyList_Check(arg)) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(arg);
    if (n == 0) Py_RETURN_NONE;
    /* walk 1: lengths, widest element, fixed length? */
    uint64_t total = 0;
    int wide = 0, fixed = 1;
    Py_ssize_t first = -1;
    for (Py_ssize_t i = 0; i < n; ++i) {
        const void *d;
        Py_ssize_t len;
        int kind;
        const int r = key_view(PyList_GET_ITEM(arg, i), &d, &len, &kind);
        if (r < 0) return NULL;
        if (r > 0) Py_RETURN_NONE;
        if (kind > 1) wide = 1;
        if (first < 0) first = len;
        else if (len != first) fixed = 0;
        total += (uint64_t)len;
    }
    const int layout = wide ? 2 : (fixed ? 0 : 1);
    const uint64_t elem = wide ? 4 : 1;
    PyObject *blob = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)(total ? total * elem : 1));
    if (!blob) return NULL;
    PyObject *offs = NULL;
    uint64_t *op = NULL;
    if (layout != 0) {
        offs = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)((uint64_t)(n + 1) * 8));
        if (!offs) {
            Py_DECREF(blob);
            return NULL;
        }
        op = (uint64_t *)PyByteArray_AS_STRING(offs);
    }
    uint8_t *bp = (uint8_t *)PyByteArray_AS_STRING(blob);
    if (total == 0) bp[0] = 0;
    /* walk 2: the elements.  (The list is ours for the duration: no python code runs between the walks -- but a bytearray could
       in principle have been resized by another thread holding no GIL-free section; the bound check keeps the copy inside the blob.) */
    uint64_t at = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        const void *d;
        Py_ssize_t len;
        int kind;
        if (key_view(PyList_GET_ITEM(arg, i), &d, &len, &kind) != 0 || at + (uint64_t)len > total || (!wide && kind > 1)) {
            Py_DECREF(blob);
            Py_XDECREF(offs);
            PyErr_SetString(PyExc_RuntimeError, "key list changed while it was packed");
            return NULL;
        }
        if (op) op[i] = at;
        if (!wide) {
            memcpy(bp + at, d, (size_t)len);
        } else {
            uint32_t *w = (uint32_t *)bp + at;
            if (kind == 1) {
                const uint8_t *s = (const uint8_t *)d;
                for (Py_ssize_t j = 0; j < len; ++j) w[j] = s[j];
            } else if (kind == 2) {
                const uint16_t *s = (const uint16_t *)d;
                for (Py_ssize_t j = 0; j < len; ++j) w[j] = s[j];
            } else {
                memcpy(w, d, (size_t)len * 4);
            }
        }
        at += (uint64_t)len;
    }
    if (op) op[n] = at;
    if (at != total) {
        Py_DECREF(blob);
        Py_XDECREF(offs);
        PyErr_SetString(PyExc_RuntimeError, "key list changed while it was packed");
        return NULL;
    }
    PyObject *res = Py_BuildValue("(iOOnn)", layout, blob, offs ? offs : Py_None, n, layout == 0 ? first : (Py_ssize_t)0);
    Py_DECREF(blob);
    Py_XDECREF(offs);
    return res;
}

static PyMethodDef methods[] = {
    {"pack", pack, METH_O, "pack(list of str / bytes / bytearray) -> (layout, blob, offsets | None, n, key_len) or None"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pylist", "host-side packer of key lists (see psk_pylist.c)", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pylist(void) { return PyModule_Create(&moddef); }

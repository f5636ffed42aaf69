// _mq_stage — host-side staging of Pillow images for the MI355X preprocessing kernels (CPython extension, host code only).
//
// The reference hands `vectorise()` PIL images (s2_inference/clip_utils.py:84-118 format_and_load_CLIP_image returns PIL.Image objects, which
// torchvision's transform then reads one at a time).  Here their pixels go to one pinned staging buffer and cross PCIe in one copy
// (engine/preprocess.py::PackedImages).  Getting at the pixels of N images used to be N trips through Python (Pillow's Arrow export +
// pyarrow import + NumPy view: 7-11 us each, all under the GIL, so concurrent request threads queued behind each other: 4 callers of 256
// images were no faster than one, profiles/r02ab_pack_chunks_ab.txt).  This module does the whole batch in ONE call:
//
//   gather_rgbx(images, dst_address, dst_capacity, offsets, nbytes, threads) -> list of the indices it could not export
//
//   * under the GIL, per image: `img.__arrow_c_array__()` (the Arrow PyCapsule interface Pillow >= 11.2 implements; it loads lazy images
//     first) -> ArrowSchema / ArrowArray capsules, checked to be the FixedSizeList<uint8>[4] of an RGB / RGBX image ("+w:4" over "C",
//     zero offsets, 4 * H * W bytes) — anything else (an image stored in several blocks, another mode, an older Pillow) is reported back by
//     index and the caller stages that image the slow way;
//   * with the GIL RELEASED: a few threads memcpy the pixel blocks to dst_address + offsets[i];
//   * under the GIL again: the capsules are dropped (their destructors release the Arrow arrays, i.e. Pillow's reference on the pixels).
//
// No HIP, no torch: plain CPython C API + the Arrow C data interface structs (a frozen ABI, declared below).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#include "copy_pool.h"

namespace {

// Arrow C data interface (https://arrow.apache.org/docs/format/CDataInterface.html — the struct layouts are frozen by the specification)
struct ArrowSchema {
    const char* format; const char* name; const char* metadata; int64_t flags; int64_t n_children;
    struct ArrowSchema** children; struct ArrowSchema* dictionary; void (*release)(struct ArrowSchema*); void* private_data;
};
struct ArrowArray {
    int64_t length; int64_t null_count; int64_t offset; int64_t n_buffers; int64_t n_children;
    const void** buffers; struct ArrowArray** children; struct ArrowArray* dictionary; void (*release)(struct ArrowArray*); void* private_data;
};

struct Item { const void* src; int64_t dst_off; int64_t bytes; };


// pixel block of one exported image, or nullptr (no Python error left pending)
const void* rgbx_block(PyObject* capsules, int64_t want_bytes) {
    if (!PyTuple_Check(capsules) || PyTuple_GET_SIZE(capsules) != 2) return nullptr;
    PyObject* sc = PyTuple_GET_ITEM(capsules, 0);
    PyObject* ac = PyTuple_GET_ITEM(capsules, 1);
    if (!PyCapsule_IsValid(sc, "arrow_schema") || !PyCapsule_IsValid(ac, "arrow_array")) return nullptr;
    const ArrowSchema* s = (const ArrowSchema*)PyCapsule_GetPointer(sc, "arrow_schema");
    const ArrowArray* a = (const ArrowArray*)PyCapsule_GetPointer(ac, "arrow_array");
    if (!s || !a || !s->release || !a->release) return nullptr;
    if (!s->format || strcmp(s->format, "+w:4") != 0 || s->n_children != 1 || !s->children || !s->children[0] || !s->children[0]->format ||
        strcmp(s->children[0]->format, "C") != 0)
        return nullptr;
    if (a->offset != 0 || a->null_count > 0 || a->n_children != 1 || !a->children || !a->children[0]) return nullptr;
    const ArrowArray* c = a->children[0];
    if (c->offset != 0 || c->null_count > 0 || c->n_buffers < 2 || !c->buffers || !c->buffers[1]) return nullptr;
    if (a->length * 4 != want_bytes || c->length != want_bytes) return nullptr;
    return c->buffers[1];
}

// the same block through the image's ImagingCore (`img.im`), for an image that is loaded and of mode "RGB"; *keep receives the capsule that
// owns Pillow's reference on the pixels.  nullptr (no Python error pending, *keep possibly set: the caller drops it) when this route does not apply.
const void* core_rgb_block(PyObject* img, PyObject* s_im, PyObject* s_mode, PyObject* method, int64_t want_bytes, PyObject** keep) {
    *keep = nullptr;
    PyObject* mode = PyObject_GetAttr(img, s_mode);
    if (!mode) { PyErr_Clear(); return nullptr; }
    const bool rgb = PyUnicode_Check(mode) && PyUnicode_CompareWithASCIIString(mode, "RGB") == 0;
    Py_DECREF(mode);
    if (!rgb) return nullptr;
    PyObject* core = PyObject_GetAttr(img, s_im);
    if (!core) { PyErr_Clear(); return nullptr; }
    const void* src = nullptr;
    if (core != Py_None) {
        PyObject* cap = PyObject_CallMethodNoArgs(core, method);
        if (!cap) PyErr_Clear();
        else if (PyCapsule_IsValid(cap, "arrow_array")) {
            const ArrowArray* a = (const ArrowArray*)PyCapsule_GetPointer(cap, "arrow_array");
            if (a && a->release && a->offset == 0 && a->null_count <= 0 && a->n_children == 1 && a->children && a->children[0] && a->length * 4 == want_bytes) {
                const ArrowArray* c = a->children[0];
                if (c->offset == 0 && c->null_count <= 0 && c->n_buffers >= 2 && c->buffers && c->buffers[1] && c->length == want_bytes) src = c->buffers[1];
            }
            *keep = cap;
        } else {
            Py_DECREF(cap);
        }
    }
    Py_DECREF(core);
    return src;
}

bool int64_view(PyObject* obj, Py_buffer* view, Py_ssize_t n, const char* what) {
    if (PyObject_GetBuffer(obj, view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return false;
    if (view->itemsize != 8 || view->len != n * 8 || (view->format && strcmp(view->format, "l") != 0 && strcmp(view->format, "q") != 0)) {
        PyBuffer_Release(view);
        PyErr_Format(PyExc_ValueError, "%s must be a C-contiguous int64 buffer of %zd items", what, n);
        return false;
    }
    return true;
}

PyObject* gather_rgbx(PyObject*, PyObject* args) {
    PyObject *images, *dst_obj, *off_obj, *len_obj;
    long long capacity = 0;
    int threads = 1;
    if (!PyArg_ParseTuple(args, "OOLOO|i", &images, &dst_obj, &capacity, &off_obj, &len_obj, &threads)) return nullptr;
    if (!PyList_Check(images)) { PyErr_SetString(PyExc_TypeError, "images must be a list"); return nullptr; }
    const Py_ssize_t n = PyList_GET_SIZE(images);
    char* dst = (char*)PyLong_AsVoidPtr(dst_obj);
    if (PyErr_Occurred()) return nullptr;
    if (!dst && n) { PyErr_SetString(PyExc_ValueError, "null destination"); return nullptr; }
    Py_buffer offs, lens;
    if (!int64_view(off_obj, &offs, n, "offsets")) return nullptr;
    if (!int64_view(len_obj, &lens, n, "nbytes")) { PyBuffer_Release(&offs); return nullptr; }
    const int64_t* off = (const int64_t*)offs.buf;
    const int64_t* len = (const int64_t*)lens.buf;
    for (Py_ssize_t i = 0; i < n; ++i)   // a slot outside the destination is the caller's bug, not an image that cannot be exported
        if (len[i] > 0 && off[i] >= 0 && (off[i] > (int64_t)capacity || len[i] > (int64_t)capacity - off[i])) {
            PyBuffer_Release(&offs);
            PyBuffer_Release(&lens);
            PyErr_Format(PyExc_ValueError, "item %zd: %lld bytes at offset %lld do not fit the %lld-byte destination", i, (long long)len[i],
                         (long long)off[i], capacity);
            return nullptr;
        }

    PyObject* failed = PyList_New(0);
    PyObject* method = PyUnicode_InternFromString("__arrow_c_array__");
    PyObject* s_im = PyUnicode_InternFromString("im");
    PyObject* s_mode = PyUnicode_InternFromString("mode");
    std::vector<PyObject*> keep;      // the capsule pairs: they own Pillow's reference on the pixel blocks until the copies are done
    std::vector<Item> items;
    keep.reserve(n);
    items.reserve(n);
    int64_t total = 0;
    bool ok = failed && method;
    for (Py_ssize_t i = 0; ok && i < n; ++i) {
        const void* src = nullptr;
        PyObject* caps = nullptr;
        if (len[i] > 0 && off[i] >= 0 && i < PyList_GET_SIZE(images)) {   // (the export runs Python code: hold the item, re-check the list)
            PyObject* img = PyList_GET_ITEM(images, i);
            Py_INCREF(img);
            // A loaded RGB image: ask its ImagingCore for the array capsule directly (Image.__arrow_c_array__ is a Python-level wrapper:
            // load() + TWO capsules, 2.4 us; the core's array capsule alone is 0.55 us; the layout of mode "RGB" is known — 4 bytes per
            // pixel — and the length is checked below).  Anything else — a lazy file image, another mode — goes through the wrapper.
            src = s_im && s_mode ? core_rgb_block(img, s_im, s_mode, method, len[i], &caps) : nullptr;
            if (!src) {
                Py_XDECREF(caps);
                caps = PyObject_CallMethodNoArgs(img, method);
                if (!caps) PyErr_Clear();                 // (several blocks, unsupported mode, no Arrow interface: the caller's slow route)
                else src = rgbx_block(caps, len[i]);
            }
            Py_DECREF(img);
        }
        if (src) {
            keep.push_back(caps);
            items.push_back({src, off[i], len[i]});
            total += len[i];
        } else {
            Py_XDECREF(caps);
            PyObject* idx = PyLong_FromSsize_t(i);
            ok = idx && PyList_Append(failed, idx) == 0;
            Py_XDECREF(idx);
        }
    }
    if (ok && !items.empty()) {
        int t = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
        if (total < (4 << 20)) t = 1;                     // small packs: waking the workers costs more than the copy
        Py_BEGIN_ALLOW_THREADS
        // No C++ exception may unwind through this CPython frame (the GIL is released: it would end in std::terminate): whatever the range setup
        // or the pool throws (bad_alloc, a system_error from a mutex) falls back to a serial copy of everything — memcpy is idempotent, ranges a
        // worker already copied are simply copied again.
        // (`ranges` and `job` live OUTSIDE the try: the pool's workers run `job` — CopyPool::run does not return, not even by an exception, while one
        // of them is inside it, and the fallback below must not find them destroyed)
        std::vector<std::pair<size_t, size_t>> ranges;
        const std::function<void(int)> job = [&](int r) {
            for (size_t k = ranges[r].first; k < ranges[r].second; ++k) memcpy(dst + items[k].dst_off, items[k].src, (size_t)items[k].bytes);
        };
        try {
            // ranges of about equal bytes, whole images each
            {
                const int64_t share = (total + t - 1) / t;
                size_t lo = 0;
                int64_t acc = 0;
                for (size_t k = 0; k < items.size(); ++k) {
                    acc += items[k].bytes;
                    if (acc >= share || k + 1 == items.size()) {
                        ranges.emplace_back(lo, k + 1);
                        lo = k + 1;
                        acc = 0;
                    }
                }
            }
            mq_copy_pool().run((int)ranges.size(), job);
        } catch (...) {
            for (size_t k = 0; k < items.size(); ++k) memcpy(dst + items[k].dst_off, items[k].src, (size_t)items[k].bytes);
        }
        Py_END_ALLOW_THREADS
    }
    for (PyObject* c : keep) Py_DECREF(c);
    Py_XDECREF(method);
    Py_XDECREF(s_im);
    Py_XDECREF(s_mode);
    PyBuffer_Release(&offs);
    PyBuffer_Release(&lens);
    if (!ok) { Py_XDECREF(failed); return PyErr_Occurred() ? nullptr : PyErr_NoMemory(); }
    return failed;
}

// rgb_sizes(images, heights, widths) -> True when EVERY item is a loaded Pillow image of mode "RGB" with a non-empty size (heights / widths:
// writable int32 buffers of len(images) items, filled); False otherwise (nothing to rely on in the buffers).  One call instead of a Python
// loop of isinstance / mode / size checks per image (~2 us each under the GIL: 0.5 ms of a 256-image request).
PyObject* rgb_sizes(PyObject*, PyObject* args) {
    PyObject *images, *h_obj, *w_obj;
    if (!PyArg_ParseTuple(args, "OOO", &images, &h_obj, &w_obj)) return nullptr;
    if (!PyList_Check(images)) Py_RETURN_FALSE;
    const Py_ssize_t n = PyList_GET_SIZE(images);
    Py_buffer hb, wb;
    if (PyObject_GetBuffer(h_obj, &hb, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    if (PyObject_GetBuffer(w_obj, &wb, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&hb); return nullptr; }
    bool ok = hb.len == n * 4 && wb.len == n * 4 && n > 0;
    PyObject* s_im = PyUnicode_InternFromString("im");
    PyObject* s_mode = PyUnicode_InternFromString("mode");
    PyObject* s_size = PyUnicode_InternFromString("size");
    PyObject* s_arrow = PyUnicode_InternFromString("__arrow_c_array__");
    ok = ok && s_im && s_mode && s_size && s_arrow;
    int32_t* hs = (int32_t*)hb.buf;
    int32_t* ws = (int32_t*)wb.buf;
    for (Py_ssize_t i = 0; ok && i < n && i < PyList_GET_SIZE(images); ++i) {
        PyObject* img = PyList_GET_ITEM(images, i);
        Py_INCREF(img);
        PyObject* mode = PyObject_GetAttr(img, s_mode);
        PyObject* core = mode ? PyObject_GetAttr(img, s_im) : nullptr;
        PyObject* size = core ? PyObject_GetAttr(img, s_size) : nullptr;
        ok = mode && core && size && PyUnicode_Check(mode) && PyUnicode_CompareWithASCIIString(mode, "RGB") == 0 && core != Py_None &&
             PyObject_HasAttr(core, s_arrow) && PyTuple_Check(size) && PyTuple_GET_SIZE(size) == 2;
        if (ok) {
            const long w = PyLong_AsLong(PyTuple_GET_ITEM(size, 0)), h = PyLong_AsLong(PyTuple_GET_ITEM(size, 1));
            ok = !PyErr_Occurred() && w > 0 && h > 0 && w < (1 << 30) && h < (1 << 30);
            if (ok) { hs[i] = (int32_t)h; ws[i] = (int32_t)w; }
        }
        PyErr_Clear();
        Py_XDECREF(mode);
        Py_XDECREF(core);
        Py_XDECREF(size);
        Py_DECREF(img);
    }
    Py_XDECREF(s_im);
    Py_XDECREF(s_mode);
    Py_XDECREF(s_size);
    Py_XDECREF(s_arrow);
    PyBuffer_Release(&hb);
    PyBuffer_Release(&wb);
    if (ok) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

PyMethodDef methods[] = {
    {"rgb_sizes", rgb_sizes, METH_VARARGS,
     "rgb_sizes(images, heights_i32, widths_i32) -> bool: every item is a loaded Pillow RGB image; its height / width were written"},
    {"gather_rgbx", gather_rgbx, METH_VARARGS,
     "gather_rgbx(images, dst_address, dst_capacity, offsets, nbytes, threads=1) -> [indices not exported]\n"
     "Copy the RGBX pixel blocks (4 bytes per pixel, Pillow's in-memory layout of RGB images) of a list of PIL images to\n"
     "dst_address + offsets[i] inside a destination of dst_capacity bytes; nbytes[i] = 4 * height * width.  The copies run\n"
     "with the GIL released."},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_mq_stage", "host-side staging of Pillow images (marqo_amd)", -1, methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__mq_stage(void) { return PyModule_Create(&module); }

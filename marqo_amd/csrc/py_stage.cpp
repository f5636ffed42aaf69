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
#include <thread>
#include <vector>

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
            caps = PyObject_CallMethodNoArgs(img, method);
            Py_DECREF(img);
            if (!caps) PyErr_Clear();                     // (several blocks, unsupported mode, no Arrow interface: the caller's slow route)
            else src = rgbx_block(caps, len[i]);
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
        if (total < (4 << 20)) t = 1;                     // small packs: a thread start costs more than the copy
        Py_BEGIN_ALLOW_THREADS
        auto work = [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) memcpy(dst + items[k].dst_off, items[k].src, (size_t)items[k].bytes);
        };
        if (t == 1) {
            work(0, items.size());
        } else {
            std::vector<std::thread> pool;
            const int64_t share = (total + t - 1) / t;
            size_t lo = 0;
            int64_t acc = 0;
            for (size_t k = 0; k < items.size(); ++k) {
                acc += items[k].bytes;
                if (acc >= share || k + 1 == items.size()) {
                    if (k + 1 == items.size() || (int)pool.size() == t - 1) { work(lo, items.size()); break; }   // the caller's thread takes the last range
                    try {
                        pool.emplace_back(work, lo, k + 1);
                    } catch (...) {                       // no thread to be had: this thread copies the rest
                        work(lo, items.size());
                        break;
                    }
                    lo = k + 1;
                    acc = 0;
                }
            }
            for (auto& th : pool) th.join();
        }
        Py_END_ALLOW_THREADS
    }
    for (PyObject* c : keep) Py_DECREF(c);
    Py_XDECREF(method);
    PyBuffer_Release(&offs);
    PyBuffer_Release(&lens);
    if (!ok) { Py_XDECREF(failed); return PyErr_Occurred() ? nullptr : PyErr_NoMemory(); }
    return failed;
}

PyMethodDef methods[] = {
    {"gather_rgbx", gather_rgbx, METH_VARARGS,
     "gather_rgbx(images, dst_address, dst_capacity, offsets, nbytes, threads=1) -> [indices not exported]\n"
     "Copy the RGBX pixel blocks (4 bytes per pixel, Pillow's in-memory layout of RGB images) of a list of PIL images to\n"
     "dst_address + offsets[i] inside a destination of dst_capacity bytes; nbytes[i] = 4 * height * width.  The copies run\n"
     "with the GIL released."},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_mq_stage", "host-side staging of Pillow images (marqo_amd)", -1, methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__mq_stage(void) { return PyModule_Create(&module); }

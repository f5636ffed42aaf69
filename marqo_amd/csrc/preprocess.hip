// Image preprocessing on the GPU (K10 / K11 of SURVEY.md §8a): Pillow-exact antialiased bicubic
// resampling of uint8 images, centre crop, grid chunking, ToTensor + Normalize.
//
// What is replaced (reference file:line):
//   * Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor -> Normalize
//         src/marqo/s2_inference/clip_utils.py:48-67, open_clip image_transform_v2 (open_clip_model.py:84-85)
//   * PatchifySimple: image.resize((240,240)) + generate_boxes + image.crop(bb)
//         src/marqo/s2_inference/processing/image.py:120-151, image_utils.py:165-202,267-279
// The reference runs these per image in Python download threads on PIL (Pillow 10.4.0); the bytes it
// produces are defined by Pillow's 8-bit two-pass resampler: per-output-pixel windows whose support is
// scaled by the downscale factor, double-precision bicubic weights normalised to 1 and rounded to 22-bit
// fixed point, horizontal pass -> round/clip to uint8 -> vertical pass -> round/clip.  We reproduce it
// bit for bit: the (tiny) coefficient tables are computed on the host in double exactly as published, the
// integer passes run on the GPU.
//
// This is byte/integer, HBM-bound work — deliberately NOT reshaped into a GEMM.  Layout:
//   H pass: one workgroup per (job, source row): the needed span of the row is staged in LDS with
//           coalesced loads, every thread produces output pixels from LDS taps, rows written as uint8
//           into the caller's workspace (only the rows / columns the crop needs are ever computed);
//   V pass: one workgroup per (job, output row); threads own byte columns, so every tap is one coalesced
//           row read of the intermediate.
#include <math.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow's triangle filter (Image.BILINEAR, support 1): CLIPA's preprocessing (open_clip _apcfg: bilinear squash)
double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

constexpr int FILTER_NEAREST = 0, FILTER_BILINEAR = 2, FILTER_BICUBIC = 3;  // Pillow's Image.NEAREST / BILINEAR / BICUBIC
double filter_support(int filter) { return filter == FILTER_BILINEAR ? 1.0 : 2.0; }

int coeff_ksize(int in_size, int out_size, int filter = FILTER_BICUBIC) {
    if (filter == FILTER_NEAREST) return 1;
    double filterscale = (double)((float)in_size - 0.0f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(filter_support(filter) * filterscale) * 2 + 1;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output positions [first, first+count).
// bounds: (xmin, n) pairs; kk: [count][ksize]
void compute_coeffs(int in_size, int out_size, int first, int count, int ksize, int32_t* bounds, int32_t* kk, int filter = FILTER_BICUBIC) {
    if (filter == FILTER_NEAREST) {
        // Image.resize(..., NEAREST) — what Pillow runs for palette ("P") and bilevel ("1") images WHATEVER filter the caller asked
        // for (Image.resize: `if self.mode in ("1", "P"): resample = NEAREST`): _imaging.c::_resize builds the affine
        // a0 = in / out and Geometry.c::ImagingScaleAffine tabulates xin = COORD(xo), xo starting at a0 * 0.5 and advanced by
        // REPEATED ADDITION of a0 (so the table is reproduced with the same accumulation, not with a product).  As a one-tap
        // "filter" of weight 1.0 it runs through the same two integer passes: clip8(2^21 + p * 2^22) == p.
        const double a0 = (double)in_size / out_size;
        double xo = a0 * 0.5;
        for (int xx = 0; xx < first + count; ++xx) {
            if (xx >= first) {
                int xin = xo < 0.0 ? 0 : (int)xo;
                if (xin > in_size - 1) xin = in_size - 1;
                bounds[2 * (xx - first)] = xin;
                bounds[2 * (xx - first) + 1] = 1;
                kk[(size_t)(xx - first) * ksize] = 1 << PRECISION_BITS;
            }
            xo += a0;
        }
        return;
    }
    const float in0 = 0.0f, in1 = (float)in_size;
    double scale, filterscale;
    filterscale = scale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = filter_support(filter) * filterscale;
    std::vector<double> k(ksize);
    for (int i = 0; i < count; ++i) {
        const int xx = first + i;
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; x++) {
            const double xa = (x + xmin - center + 0.5) * ss;
            const double w = filter == FILTER_BILINEAR ? bilinear_filter(xa) : bicubic_filter(xa);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        bounds[2 * i] = xmin;
        bounds[2 * i + 1] = xmax;
        int32_t* o = kk + (size_t)i * ksize;
        for (x = 0; x < ksize; x++)
            o[x] = k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int32_t)(0.5 + k[x] * (1 << PRECISION_BITS));
    }
}

// One resample+crop job (device-visible POD).  Offsets are in BYTES for pixels, in int32 ELEMENTS for
// coefficient tables (relative to the coefficient pool).
struct Job {
    int64_t src_off;      // first byte of the source sub-image
    int64_t dst_off;      // first byte of the destination region
    int64_t tmp_off;      // first byte of this job's intermediate rows (workspace)
    int64_t hb, hk;       // horizontal bounds / weights  (-1: identity axis, crop start in h_first)
    int64_t vb, vk;       // vertical bounds / weights    (-1: identity axis, crop start in v_first)
    int32_t src_stride, dst_stride;
    int32_t in_w, in_h;   // source sub-image size
    int32_t out_w, out_h; // written region (after the crop)
    int32_t h_first, v_first;  // first kept output column / row of the uncropped resize (identity axis: source col/row)
    int32_t ksize_h, ksize_v;
    int32_t row0, rows;   // source rows the vertical pass needs: [row0, row0 + rows)
    int32_t col0, cols;   // source columns the horizontal pass needs: [col0, col0 + cols)
};

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Pillow's premultiplied-alpha round trip around the resize of RGBA / LA images (Image.resize converts RGBA -> "RGBa", resamples all four
// bands, converts back; Convert.c): premultiply = MULDIV255(c, a), un-premultiply = c if a is 0 or 255 else CLIP8(255 * c / a).
__device__ __forceinline__ uint32_t premultiply_rgba(uint32_t px) {
    const uint32_t a = px >> 24;
    uint32_t out = px & 0xff000000u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t t = ((px >> (8 * c)) & 0xffu) * a + 128u;
        out |= (((t >> 8) + t) >> 8) << (8 * c);
    }
    return out;
}
__device__ __forceinline__ uint8_t unpremultiply(int c, int a) {
    if (a == 255 || a == 0) return (uint8_t)c;
    const int v = (255 * c) / a;
    return (uint8_t)(v > 255 ? 255 : v);
}

// ---- horizontal pass: tmp[r][x][c], r in [0, rows), x in [0, out_w) ---------------------------------
// C = bytes per pixel: 3 (RGB) or 4 (RGBA sources: premultiplied while the row is staged in LDS, all four bands resampled)
template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                        const Job* __restrict__ jobs, const int32_t* __restrict__ coeffs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srow[];
    const Job j = jobs[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= j.rows) return;
    const uint8_t* in = src + j.src_off + (int64_t)(j.row0 + r) * j.src_stride + (int64_t)j.col0 * C;
    if (C == 4) {  // (rows are 4-byte aligned: images start on 256-byte boundaries and a row is 4 * w bytes)
        // an image that is not resized at all (both axes identity: torchvision's Resize returns it untouched, Image.resize copies) never
        // takes the premultiplied round trip, which is lossy — its colour bytes pass through as they are
        const bool raw = j.hb < 0 && j.vb < 0;
        for (int i = threadIdx.x; i < j.cols; i += 256) {
            const uint32_t px = ((const uint32_t*)in)[i];
            ((uint32_t*)srow)[i] = raw ? px : premultiply_rgba(px);
        }
    } else {
        const int nbytes = j.cols * C;
        for (int i = threadIdx.x; i < nbytes; i += 256) srow[i] = in[i];
    }
    __syncthreads();
    uint8_t* out = tmp + j.tmp_off + (int64_t)r * j.out_w * C;
    if (j.hb < 0) {  // identity axis: plain crop copy
        for (int i = threadIdx.x; i < j.out_w * C; i += 256) out[i] = srow[i];
        return;
    }
    const int32_t* bounds = coeffs + j.hb;
    const int32_t* kk = coeffs + j.hk;
    for (int x = threadIdx.x; x < j.out_w; x += 256) {
        const int xmin = bounds[2 * x] - j.col0, n = bounds[2 * x + 1];
        const int32_t* k = kk + (int64_t)x * j.ksize_h;
        int s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 1 << (PRECISION_BITS - 1);
        const uint8_t* p = srow + xmin * C;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += (int)p[C * t + c] * w;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[C * x + c] = clip8(s[c]);
    }
}

// ---- vertical pass: dst[y][x*3+c] -------------------------------------------------------------------------
// C = 3: a thread owns a byte column.  C = 4: a thread owns a pixel (it needs the pixel's resampled alpha to un-premultiply) and writes
// its three colour bytes — the `.convert("RGB")` that follows in the reference's transform only drops the alpha band.
template <int C>
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                        const Job* __restrict__ jobs, const int32_t* __restrict__ coeffs) {
    const Job j = jobs[blockIdx.y];
    const int y = blockIdx.x;
    if (y >= j.out_h) return;
    const int rowbytes = j.out_w * C;
    const uint8_t* in = tmp + j.tmp_off;
    uint8_t* out = dst + j.dst_off + (int64_t)y * j.dst_stride;
    if (C == 4) {
        const bool ident = j.vb < 0;
        const int ymin = ident ? y : (coeffs + j.vb)[2 * y] - j.row0, n = ident ? 1 : (coeffs + j.vb)[2 * y + 1];
        const int32_t* k = ident ? nullptr : coeffs + j.vk + (int64_t)y * j.ksize_v;
        for (int x = threadIdx.x; x < j.out_w; x += 256) {
            const uint8_t* p = in + (int64_t)ymin * rowbytes + x * 4;
            int v[4];
            if (ident) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = p[c];
            } else {
                int s[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
                for (int t = 0; t < n; ++t) {
                    const uint32_t q = *(const uint32_t*)(p + (int64_t)t * rowbytes);
                    const int w = k[t];
#pragma unroll
                    for (int c = 0; c < 4; ++c) s[c] += (int)((q >> (8 * c)) & 0xffu) * w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = clip8(s[c]);
            }
            const int a = (j.hb < 0 && j.vb < 0) ? 255 : v[3];   // (un-resized image: colour bytes as they are, see the H pass)
#pragma unroll
            for (int c = 0; c < 3; ++c) out[3 * x + c] = unpremultiply(v[c], a);
        }
        return;
    }
    if (j.vb < 0) {
        const uint8_t* p = in + (int64_t)y * rowbytes;  // rows were already restricted to the crop
        for (int i = threadIdx.x; i < rowbytes; i += 256) out[i] = p[i];
        return;
    }
    const int ymin = (coeffs + j.vb)[2 * y] - j.row0, n = (coeffs + j.vb)[2 * y + 1];
    const int32_t* k = coeffs + j.vk + (int64_t)y * j.ksize_v;
    for (int i = threadIdx.x; i < rowbytes; i += 256) {
        int s = 1 << (PRECISION_BITS - 1);
        const uint8_t* p = in + (int64_t)ymin * rowbytes + i;
        for (int t = 0; t < n; ++t) s += (int)p[(int64_t)t * rowbytes] * k[t];
        out[i] = clip8(s);
    }
}

// ---- ToTensor + Normalize: uint8 [n,S,S,3] -> fp32 [n,3,S,S] --------------------------------------------------
__global__ __launch_bounds__(256) void to_tensor_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t npix_total,
                                                       int64_t plane, float m0, float m1, float m2, float s0, float s1, float s2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix_total; i += (int64_t)gridDim.x * 256) {
        const int64_t img = i / plane, p = i - img * plane;
        const uint8_t* px = in + i * 3;
        float* o = out + img * 3 * plane + p;
        // torchvision: img.float().div(255) then (x - mean) / std   (clip_utils.py:65-66)
        o[0] = ((float)px[0] / 255.0f - m0) / s0;
        o[plane] = ((float)px[1] / 255.0f - m1) / s1;
        o[2 * plane] = ((float)px[2] / 255.0f - m2) / s2;
    }
}

// ---- host-side planning ----------------------------------------------------------------------------------------
struct Plan {
    std::vector<Job> jobs;
    std::vector<int32_t> coeffs;
    size_t tmp_bytes = 0;
    int max_rows = 0, max_out_h = 0, max_cols = 0;
    int filter = FILTER_BICUBIC;
    int ch = 3;   // source bytes per pixel: 3 (RGB), 4 (RGBA: premultiplied resize of all four bands, RGB written)

    // add a job: source sub-image (in_h x in_w) resized to (rs_h x rs_w), keep [top, top+out_h) x [left, left+out_w)
    void add(int64_t src_off, int src_stride, int in_h, int in_w, int rs_h, int rs_w, int top, int left, int out_h, int out_w,
             int64_t dst_off, int dst_stride) {
        Job j;
        memset(&j, 0, sizeof(j));
        j.src_off = src_off; j.src_stride = src_stride; j.in_h = in_h; j.in_w = in_w;
        j.dst_off = dst_off; j.dst_stride = dst_stride; j.out_h = out_h; j.out_w = out_w;
        // horizontal axis
        if (rs_w == in_w) {
            j.hb = j.hk = -1; j.h_first = left; j.col0 = left; j.cols = out_w; j.ksize_h = 0;
        } else {
            j.ksize_h = coeff_ksize(in_w, rs_w, filter);
            j.hb = (int64_t)coeffs.size(); coeffs.resize(coeffs.size() + 2 * (size_t)out_w);
            j.hk = (int64_t)coeffs.size(); coeffs.resize(coeffs.size() + (size_t)out_w * j.ksize_h);
            compute_coeffs(in_w, rs_w, left, out_w, j.ksize_h, coeffs.data() + j.hb, coeffs.data() + j.hk, filter);
            int lo = in_w, hi = 0;
            for (int x = 0; x < out_w; ++x) {
                const int a = coeffs[j.hb + 2 * x], b = a + coeffs[j.hb + 2 * x + 1];
                if (a < lo) lo = a;
                if (b > hi) hi = b;
            }
            j.h_first = left; j.col0 = lo; j.cols = hi - lo;
        }
        // vertical axis
        if (rs_h == in_h) {
            j.vb = j.vk = -1; j.v_first = top; j.row0 = top; j.rows = out_h; j.ksize_v = 0;
        } else {
            j.ksize_v = coeff_ksize(in_h, rs_h, filter);
            j.vb = (int64_t)coeffs.size(); coeffs.resize(coeffs.size() + 2 * (size_t)out_h);
            j.vk = (int64_t)coeffs.size(); coeffs.resize(coeffs.size() + (size_t)out_h * j.ksize_v);
            compute_coeffs(in_h, rs_h, top, out_h, j.ksize_v, coeffs.data() + j.vb, coeffs.data() + j.vk, filter);
            int lo = in_h, hi = 0;
            for (int y = 0; y < out_h; ++y) {
                const int a = coeffs[j.vb + 2 * y], b = a + coeffs[j.vb + 2 * y + 1];
                if (a < lo) lo = a;
                if (b > hi) hi = b;
            }
            j.v_first = top; j.row0 = lo; j.rows = hi - lo;
        }
        j.tmp_off = (int64_t)tmp_bytes;
        tmp_bytes = align_up(tmp_bytes + (size_t)j.rows * out_w * ch, 256);
        if (j.rows > max_rows) max_rows = j.rows;
        if (j.out_h > max_out_h) max_out_h = j.out_h;
        if (j.cols > max_cols) max_cols = j.cols;
        jobs.push_back(j);
    }
    size_t table_bytes() const { return align_up(jobs.size() * sizeof(Job), 256) + align_up(coeffs.size() * 4 + 4, 256); }
    size_t total_bytes() const { return table_bytes() + tmp_bytes; }
};

constexpr int MAX_LDS_ROW = 150 * 1024;

// upload tables and run both passes.  ws layout: [jobs][coeffs][tmp rows]
int run_plan(const Plan& p, const uint8_t* d_src, uint8_t* d_dst, void* ws, size_t ws_bytes, hipStream_t s, const char* who) {
    if (p.jobs.empty()) return MQ_OK;
    if (ws_bytes < p.total_bytes()) { mq_set_error("%s: workspace %zu < required %zu", who, ws_bytes, p.total_bytes()); return MQ_ERR_WORKSPACE; }
    MQ_CHECK_ARG(p.max_cols * p.ch <= MAX_LDS_ROW, "%s: source row span of %d pixels does not fit in LDS (max %d)", who, p.max_cols, MAX_LDS_ROW / p.ch);
    char* base = (char*)ws;
    Job* d_jobs = (Job*)base;
    int32_t* d_coeffs = (int32_t*)(base + align_up(p.jobs.size() * sizeof(Job), 256));
    uint8_t* d_tmp = (uint8_t*)(base + p.table_bytes());
    // pageable-host async copies are staged by the runtime before returning, so the vectors may die after this call
    if (hipMemcpyAsync(d_jobs, p.jobs.data(), p.jobs.size() * sizeof(Job), hipMemcpyHostToDevice, s) != hipSuccess ||
        (!p.coeffs.empty() && hipMemcpyAsync(d_coeffs, p.coeffs.data(), p.coeffs.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess)) {
        mq_set_error("%s: table upload failed: %s", who, hipGetErrorString(hipGetLastError()));
        return MQ_ERR_HIP;
    }
    MqProfScope prof(5, s);
    const size_t lds = align_up((size_t)p.max_cols * p.ch, 16);
    const void* hk = p.ch == 4 ? (const void*)resample_h_kernel<4> : (const void*)resample_h_kernel<3>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(hk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            mq_set_error("%s: hipFuncSetAttribute failed", who);
            return MQ_ERR_HIP;
        }
    }
    const size_t nj = p.jobs.size();
    for (size_t j0 = 0; j0 < nj; j0 += 65535) {  // gridDim.y limit
        const unsigned cnt = (unsigned)(nj - j0 < 65535 ? nj - j0 : 65535);
        if (p.ch == 4) {
            hipLaunchKernelGGL(resample_h_kernel<4>, dim3((unsigned)p.max_rows, cnt), dim3(256), lds, s, d_src, d_tmp, d_jobs + j0, d_coeffs);
            hipLaunchKernelGGL(resample_v_kernel<4>, dim3((unsigned)p.max_out_h, cnt), dim3(256), 0, s, d_tmp, d_dst, d_jobs + j0, d_coeffs);
        } else {
            hipLaunchKernelGGL(resample_h_kernel<3>, dim3((unsigned)p.max_rows, cnt), dim3(256), lds, s, d_src, d_tmp, d_jobs + j0, d_coeffs);
            hipLaunchKernelGGL(resample_v_kernel<3>, dim3((unsigned)p.max_out_h, cnt), dim3(256), 0, s, d_tmp, d_dst, d_jobs + j0, d_coeffs);
        }
    }
    MQ_CHECK_LAUNCH(who);
    return MQ_OK;
}

// torchvision 0.13 Resize(int): shorter side -> S, other side int(S * long / short)
void resize_output_size(int h, int w, int S, int* nh, int* nw) {
    const int shrt = w <= h ? w : h, lng = w <= h ? h : w;
    if (shrt == S) { *nh = h; *nw = w; return; }
    const int new_long = (int)((double)S * (double)lng / (double)shrt);
    if (w <= h) { *nw = S; *nh = new_long; } else { *nh = S; *nw = new_long; }
}
// CenterCrop offsets: int(round((dim - S) / 2.0)) with Python's round-half-to-even
int center_off(int dim, int S) {
    const int d = dim - S;
    if (d % 2 == 0) return d / 2;
    const int fl = (d - 1) / 2;           // d odd and >= 1: value fl + 0.5 -> the even neighbour
    return (fl % 2 == 0) ? fl : fl + 1;
}

void add_clip_job(Plan& p, int64_t src_off, int src_stride, int h, int w, int S, int64_t dst_off) {
    int nh, nw;
    resize_output_size(h, w, S, &nh, &nw);
    p.add(src_off, src_stride, h, w, nh, nw, center_off(nh, S), center_off(nw, S), S, S, dst_off, S * 3);
}

// generate_boxes (image_utils.py:165-202) on the 240x240 working image, preceded by the whole image
void grid_boxes(int size, int hn, int wn, int overlap, std::vector<int>& boxes) {
    boxes.insert(boxes.end(), {0, 0, size, size});
    const int height = size / hn, width = size / wn;
    for (int i = 0; i < size; i += height)
        for (int j = 0; j < size; j += width) {
            const int p1 = j + width, p2 = i + height;
            if (p1 > size || p2 > size) continue;
            boxes.insert(boxes.end(), {j, i, p1, p2});
            if (overlap) {
                const int p3 = p1 + width / 2, p4 = p2 + height / 2;
                if (p3 > size || p4 > size) continue;
                boxes.insert(boxes.end(), {j + width / 2, i + height / 2, p3, p4});
            }
        }
}

constexpr int CHUNK_SIZE = 240;  // image_utils.py:16-22

int check_images(const char* who, const int64_t* off, const int32_t* hs, const int32_t* ws, int64_t n) {
    MQ_CHECK_ARG(off && hs && ws, "%s: null image table", who);
    for (int64_t i = 0; i < n; ++i)
        MQ_CHECK_ARG(hs[i] >= 1 && ws[i] >= 1 && hs[i] <= 65535 && ws[i] <= 65535 && off[i] >= 0, "%s: image %ld has bad size %dx%d", who, (long)i, hs[i], ws[i]);
    return MQ_OK;
}

}  // namespace

extern "C" int mq_resample_ksize(int32_t in_size, int32_t out_size) {
    if (in_size < 1 || out_size < 1) return 0;
    return coeff_ksize(in_size, out_size);
}

extern "C" int mq_resample_coeffs(int32_t in_size, int32_t out_size, int32_t first, int32_t count, int32_t* h_bounds, int32_t* h_kk) {
    MQ_CHECK_ARG(in_size >= 1 && out_size >= 1 && first >= 0 && count >= 0 && first + count <= out_size && h_bounds && h_kk,
                 "mq_resample_coeffs: bad argument");
    compute_coeffs(in_size, out_size, first, count, coeff_ksize(in_size, out_size), h_bounds, h_kk);
    return MQ_OK;
}

extern "C" size_t mq_clip_resize_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t S) {
    if (!h_heights || !h_widths || n <= 0 || S < 1) return 0;
    Plan p;
    for (int64_t i = 0; i < n; ++i) {
        if (h_heights[i] < 1 || h_widths[i] < 1) return 0;
        add_clip_job(p, 0, h_widths[i] * 3, h_heights[i], h_widths[i], S, 0);
    }
    return p.total_bytes();
}

extern "C" int mq_clip_resize_crop_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths,
                                      int64_t n, int32_t S, uint8_t* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(S >= 1, "mq_clip_resize_crop_u8: bad output size %d", S);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_src && d_out && d_workspace, "mq_clip_resize_crop_u8: null pointer");
    MQ_TRY(check_images("mq_clip_resize_crop_u8", h_src_off, h_heights, h_widths, n));
    Plan p;
    for (int64_t i = 0; i < n; ++i) add_clip_job(p, h_src_off[i], h_widths[i] * 3, h_heights[i], h_widths[i], S, i * (int64_t)S * S * 3);
    return run_plan(p, d_src, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, "mq_clip_resize_crop_u8");
}

extern "C" size_t mq_resize_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t out_h, int32_t out_w) {
    if (!h_heights || !h_widths || n <= 0 || out_h < 1 || out_w < 1) return 0;
    Plan p;
    for (int64_t i = 0; i < n; ++i) {
        if (h_heights[i] < 1 || h_widths[i] < 1) return 0;
        p.add(0, h_widths[i] * 3, h_heights[i], h_widths[i], out_h, out_w, 0, 0, out_h, out_w, 0, out_w * 3);
    }
    return p.total_bytes();
}

extern "C" int mq_resize_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths, int64_t n,
                            int32_t out_h, int32_t out_w, uint8_t* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(out_h >= 1 && out_w >= 1, "mq_resize_u8: bad output size %dx%d", out_h, out_w);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_src && d_out && d_workspace, "mq_resize_u8: null pointer");
    MQ_TRY(check_images("mq_resize_u8", h_src_off, h_heights, h_widths, n));
    Plan p;
    for (int64_t i = 0; i < n; ++i)
        p.add(h_src_off[i], h_widths[i] * 3, h_heights[i], h_widths[i], out_h, out_w, 0, 0, out_h, out_w, i * (int64_t)out_h * out_w * 3, out_w * 3);
    return run_plan(p, d_src, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, "mq_resize_u8");
}

// PIL.Image.resize((out_w, out_h), resample=filter) with filter = 2 (Image.BILINEAR) or 3 (Image.BICUBIC): CLIPA's preprocessing is a
// bilinear squash (open_clip _apcfg(), selected by the reference at open_clip_model.py:87-97)
extern "C" size_t mq_resize_filter_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t out_h, int32_t out_w,
                                                   int32_t filter) {
    if (!h_heights || !h_widths || n <= 0 || out_h < 1 || out_w < 1 || (filter != FILTER_BILINEAR && filter != FILTER_BICUBIC)) return 0;
    Plan p;
    p.filter = filter;
    for (int64_t i = 0; i < n; ++i) {
        if (h_heights[i] < 1 || h_widths[i] < 1) return 0;
        p.add(0, h_widths[i] * 3, h_heights[i], h_widths[i], out_h, out_w, 0, 0, out_h, out_w, 0, out_w * 3);
    }
    return p.total_bytes();
}

extern "C" int mq_resize_filter_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths, int64_t n,
                                   int32_t out_h, int32_t out_w, int32_t filter, uint8_t* d_out, void* d_workspace, size_t workspace_bytes,
                                   void* stream) {
    MQ_CHECK_ARG(out_h >= 1 && out_w >= 1, "mq_resize_filter_u8: bad output size %dx%d", out_h, out_w);
    MQ_CHECK_ARG(filter == FILTER_BILINEAR || filter == FILTER_BICUBIC, "mq_resize_filter_u8: filter %d (2 = bilinear, 3 = bicubic)", filter);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_src && d_out && d_workspace, "mq_resize_filter_u8: null pointer");
    MQ_TRY(check_images("mq_resize_filter_u8", h_src_off, h_heights, h_widths, n));
    Plan p;
    p.filter = filter;
    for (int64_t i = 0; i < n; ++i)
        p.add(h_src_off[i], h_widths[i] * 3, h_heights[i], h_widths[i], out_h, out_w, 0, 0, out_h, out_w, i * (int64_t)out_h * out_w * 3, out_w * 3);
    return run_plan(p, d_src, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, "mq_resize_filter_u8");
}

// Resize by source image MODE, as Pillow itself resizes (the reference's transform calls Image.resize on whatever mode the decoder
// produced and converts to RGB afterwards: clip_utils.py:61-64, open_clip image_transform):
//   MQ_IMG_RGB (0):     uint8 [h, w, 3], the given filter (2 = bilinear, 3 = bicubic)
//   MQ_IMG_NEAREST (1): uint8 [h, w, 3] that came out of a palette ("P") or bilevel ("1") image: Pillow forces NEAREST for these modes,
//                       and nearest sampling commutes with the palette lookup, so the caller converts to RGB first
//   MQ_IMG_RGBA (2):    uint8 [h, w, 4] (RGBA; LA expanded by the caller): premultiply, resample all four bands, un-premultiply; RGB out
// crop != 0: the CLIP transform, Resize(out_h) on the shorter side + CenterCrop(out_h) (out_w must equal out_h); crop == 0: plain
// Image.resize((out_w, out_h)) ("squash").  d_out is uint8 [n, out_h, out_w, 3] in every mode.
namespace {
int mode_plan(Plan& p, const char* who, const int64_t* off, const int32_t* hs, const int32_t* ws, int64_t n, int out_h, int out_w, int filter,
              int crop, int mode) {
    MQ_CHECK_ARG(out_h >= 1 && out_w >= 1 && (!crop || out_h == out_w), "%s: bad output size %dx%d (crop needs a square)", who, out_h, out_w);
    MQ_CHECK_ARG(filter == FILTER_BILINEAR || filter == FILTER_BICUBIC, "%s: filter %d (2 = bilinear, 3 = bicubic)", who, filter);
    MQ_CHECK_ARG(mode >= 0 && mode <= 2, "%s: mode %d (0 = RGB, 1 = palette / bilevel source, 2 = RGBA)", who, mode);
    p.filter = mode == 1 ? FILTER_NEAREST : filter;
    p.ch = mode == 2 ? 4 : 3;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t so = off ? off[i] : 0, dof = off ? i * (int64_t)out_h * out_w * 3 : 0;
        if (crop) add_clip_job(p, so, ws[i] * p.ch, hs[i], ws[i], out_h, dof);
        else p.add(so, ws[i] * p.ch, hs[i], ws[i], out_h, out_w, 0, 0, out_h, out_w, dof, out_w * 3);
    }
    return MQ_OK;
}
}  // namespace

extern "C" size_t mq_resize_mode_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t out_h, int32_t out_w,
                                                 int32_t filter, int32_t crop, int32_t mode) {
    if (!h_heights || !h_widths || n <= 0) return 0;
    for (int64_t i = 0; i < n; ++i)
        if (h_heights[i] < 1 || h_widths[i] < 1) return 0;
    Plan p;
    if (mode_plan(p, "mq_resize_mode_workspace_bytes", nullptr, h_heights, h_widths, n, out_h, out_w, filter, crop, mode) != MQ_OK) return 0;
    return p.total_bytes();
}

extern "C" int mq_resize_mode_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths, int64_t n,
                                 int32_t out_h, int32_t out_w, int32_t filter, int32_t crop, int32_t mode, uint8_t* d_out, void* d_workspace,
                                 size_t workspace_bytes, void* stream) {
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_src && d_out && d_workspace, "mq_resize_mode_u8: null pointer");
    MQ_TRY(check_images("mq_resize_mode_u8", h_src_off, h_heights, h_widths, n));
    if (mode == 2)
        for (int64_t i = 0; i < n; ++i) MQ_CHECK_ARG(h_src_off[i] % 4 == 0, "mq_resize_mode_u8: RGBA image %ld must start on a 4-byte boundary", (long)i);
    Plan p;
    MQ_TRY(mode_plan(p, "mq_resize_mode_u8", h_src_off, h_heights, h_widths, n, out_h, out_w, filter, crop, mode));
    return run_plan(p, d_src, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, "mq_resize_mode_u8");
}

extern "C" int mq_chunk_grid_count(int32_t hn, int32_t wn, int32_t overlap) {
    if (hn < 1 || wn < 1 || hn > CHUNK_SIZE || wn > CHUNK_SIZE) return 0;
    std::vector<int> b;
    grid_boxes(CHUNK_SIZE, hn, wn, overlap, b);
    return (int)(b.size() / 4);
}

extern "C" size_t mq_chunk_grid_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t hn, int32_t wn,
                                                int32_t overlap, int32_t S) {
    const int nb = mq_chunk_grid_count(hn, wn, overlap);
    if (!h_heights || !h_widths || n <= 0 || S < 1 || nb == 0) return 0;
    Plan a, b;
    std::vector<int> boxes;
    grid_boxes(CHUNK_SIZE, hn, wn, overlap, boxes);
    for (int64_t i = 0; i < n; ++i) {
        if (h_heights[i] < 1 || h_widths[i] < 1) return 0;
        a.add(0, h_widths[i] * 3, h_heights[i], h_widths[i], CHUNK_SIZE, CHUNK_SIZE, 0, 0, CHUNK_SIZE, CHUNK_SIZE, 0, CHUNK_SIZE * 3);
        for (int k = 0; k < nb; ++k) add_clip_job(b, 0, CHUNK_SIZE * 3, boxes[4 * k + 3] - boxes[4 * k + 1], boxes[4 * k + 2] - boxes[4 * k], S, 0);
    }
    const size_t work = align_up((size_t)n * CHUNK_SIZE * CHUNK_SIZE * 3, 256);
    const size_t pa = a.total_bytes(), pb = b.total_bytes();
    return work + (pa > pb ? pa : pb);
}

extern "C" int mq_chunk_grid_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths,
                                int64_t n, int32_t hn, int32_t wn, int32_t overlap, int32_t S, uint8_t* d_out, float* h_boxes,
                                void* d_workspace, size_t workspace_bytes, void* stream) {
    const int nb = mq_chunk_grid_count(hn, wn, overlap);
    MQ_CHECK_ARG(nb > 0 && S >= 1, "mq_chunk_grid_u8: bad grid %dx%d / size %d", hn, wn, S);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_src && d_out && d_workspace, "mq_chunk_grid_u8: null pointer");
    MQ_TRY(check_images("mq_chunk_grid_u8", h_src_off, h_heights, h_widths, n));
    std::vector<int> boxes;
    grid_boxes(CHUNK_SIZE, hn, wn, overlap, boxes);
    const size_t work = align_up((size_t)n * CHUNK_SIZE * CHUNK_SIZE * 3, 256);
    MQ_CHECK_ARG(workspace_bytes > work, "mq_chunk_grid_u8: workspace too small");
    uint8_t* d_work = (uint8_t*)d_workspace;  // [n, 240, 240, 3]
    char* rest = (char*)d_workspace + work;
    hipStream_t s = (hipStream_t)stream;
    Plan a, b;
    for (int64_t i = 0; i < n; ++i) {
        // image.resize((240, 240)): both axes forced, no aspect preservation (image.py:143)
        a.add(h_src_off[i], h_widths[i] * 3, h_heights[i], h_widths[i], CHUNK_SIZE, CHUNK_SIZE, 0, 0, CHUNK_SIZE, CHUNK_SIZE,
              i * (int64_t)CHUNK_SIZE * CHUNK_SIZE * 3, CHUNK_SIZE * 3);
        for (int k = 0; k < nb; ++k) {
            const int x1 = boxes[4 * k], y1 = boxes[4 * k + 1], x2 = boxes[4 * k + 2], y2 = boxes[4 * k + 3];
            // image.crop(bb) then the CLIP transform of the crop
            add_clip_job(b, i * (int64_t)CHUNK_SIZE * CHUNK_SIZE * 3 + ((int64_t)y1 * CHUNK_SIZE + x1) * 3, CHUNK_SIZE * 3, y2 - y1, x2 - x1, S,
                         (i * nb + k) * (int64_t)S * S * 3);
            if (h_boxes) {  // rescale_box to original pixel coordinates (image_utils.py:141-163), as floats
                float* o = h_boxes + (i * nb + k) * 4;
                const double fx = (double)h_widths[i] / CHUNK_SIZE, fy = (double)h_heights[i] / CHUNK_SIZE;
                o[0] = (float)(x1 * fx); o[1] = (float)(y1 * fy); o[2] = (float)(x2 * fx); o[3] = (float)(y2 * fy);
            }
        }
    }
    MQ_TRY(run_plan(a, d_src, d_work, rest, workspace_bytes - work, s, "mq_chunk_grid_u8(resize)"));
    MQ_TRY(run_plan(b, d_work, d_out, rest, workspace_bytes - work, s, "mq_chunk_grid_u8(crops)"));
    return MQ_OK;
}

// ---- RGBX -> RGB: decoded Pillow images keep 4 bytes per pixel (R, G, B, pad); they are staged as they are and packed here -------
struct UnpackJob { int64_t src_off, dst_off, npix; };   // byte offsets into the staging / packed buffers

__global__ __launch_bounds__(256) void unpack_rgbx_kernel(const uint8_t* __restrict__ staging, const UnpackJob* __restrict__ jobs, uint8_t* __restrict__ dst) {
    const UnpackJob j = jobs[blockIdx.y];
    const uint8_t* in = staging + j.src_off;
    uint8_t* out = dst + j.dst_off;
    const int64_t quads = j.npix >> 2;   // 4 pixels: 16 bytes in, 12 bytes out (both offsets are 256-byte aligned)
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const uint4 v = *(const uint4*)(in + q * 16);
        uint32_t* o = (uint32_t*)(out + q * 12);
        o[0] = (v.x & 0x00ffffffu) | (v.y << 24);
        o[1] = ((v.y >> 8) & 0x0000ffffu) | (v.z << 16);
        o[2] = ((v.z >> 16) & 0x000000ffu) | (v.w << 8);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(j.npix & 3)) {
        const int64_t px = (quads << 2) + threadIdx.x;
        out[px * 3] = in[px * 4]; out[px * 3 + 1] = in[px * 4 + 1]; out[px * 3 + 2] = in[px * 4 + 2];
    }
}

extern "C" int mq_unpack_rgbx(const uint8_t* d_staging, int64_t jobs_off, int64_t n, int64_t max_npix, uint8_t* d_rgb, void* stream) {
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_staging && d_rgb && jobs_off >= 0 && jobs_off % 8 == 0 && max_npix >= 1, "mq_unpack_rgbx: bad argument");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(5, s);
    const int64_t bx = cdiv64(cdiv64(max_npix, 4), 256);
    hipLaunchKernelGGL(unpack_rgbx_kernel, dim3((unsigned)(bx < 64 ? bx : 64), (unsigned)n), dim3(256), 0, s, d_staging,
                       (const UnpackJob*)(d_staging + jobs_off), d_rgb);
    MQ_CHECK_LAUNCH("mq_unpack_rgbx");
    return MQ_OK;
}

extern "C" int mq_to_tensor_normalize(const uint8_t* d_u8, float* d_out, int64_t n, int32_t S, const float* mean, const float* std, void* stream) {
    MQ_CHECK_ARG(S >= 1 && mean && std, "mq_to_tensor_normalize: bad argument");
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_u8 && d_out, "mq_to_tensor_normalize: null pointer");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(5, s);
    const int64_t plane = (int64_t)S * S, total = n * plane;
    const unsigned grid = (unsigned)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
    hipLaunchKernelGGL(to_tensor_kernel, dim3(grid), dim3(256), 0, s, d_u8, d_out, total, plane, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    MQ_CHECK_LAUNCH("mq_to_tensor_normalize");
    return MQ_OK;
}

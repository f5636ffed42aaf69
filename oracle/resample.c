/*
 * ORACLE — test infrastructure only (never linked into libmarqo_hip.so, never imported by marqo_amd/ files).
 *
 * CPU restatement of the image resampler behind the reference's image preprocessing:
 *   - CLIP transform  Resize(n_px, BICUBIC) -> CenterCrop -> ToTensor -> Normalize
 *       src/marqo/s2_inference/clip_utils.py:48-67 (and open_clip 2.24.0 image_transform_v2, same ops)
 *   - chunker working image  image.resize((240, 240))      src/marqo/s2_inference/processing/image.py:143
 *   - patch crops            image.crop(bb)                 src/marqo/s2_inference/processing/image_utils.py:267-279
 * The arithmetic lives in a third-party dependency that is NOT in /root/reference: Pillow (pinned
 * Pillow==10.4.0 in requirements.dev.txt:29; torchvision 0.13.1 Resize on a PIL image calls
 * PIL.Image.resize(size, BICUBIC)).  This file restates Pillow's published 8-bit two-pass
 * "ImagingResample" algorithm (libImaging/Resample.c): per-output-pixel filter windows with the support
 * scaled by the downscale factor (antialiasing), double-precision coefficients normalised to sum 1, then
 * rounded to 22-bit fixed point; horizontal pass first, result rounded and clipped to uint8, then the
 * vertical pass on that uint8 intermediate.
 *
 * PINNED: tests/test_oracle_resample.py checks this restatement bit-for-bit against the Pillow installed
 * in the container (12.2.0, same algorithm) over random sizes and against committed golden fixtures
 * (tests/golden/resample_*.npz made by tests/golden/make_golden_resample.py with PIL itself).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION_BITS (32 - 8 - 2)

#define ORC_FILTER_BILINEAR 2
#define ORC_FILTER_BICUBIC 3

static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

static double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

static uint8_t clip8(int in) {
    int v = in >> PRECISION_BITS; /* arithmetic shift, as in Pillow's lookup indexing */
    if (v < 0) return 0;
    if (v > 255) return 255;
    return (uint8_t)v;
}

/* Coefficients for one axis.  Returns ksize; *bounds_out = int[2*out_size] (xmin, count),
 * *kk_out = int32[out_size * ksize] fixed-point weights.  Caller frees both. */
int orc_precompute_coeffs(int in_size, float in0, float in1, int out_size, int filter, int** bounds_out, int32_t** kk_out) {
    double (*filt)(double) = filter == ORC_FILTER_BILINEAR ? bilinear_filter : bicubic_filter;
    const double fsupport = filter == ORC_FILTER_BILINEAR ? 1.0 : 2.0;
    double scale, filterscale, support;
    filterscale = scale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    support = fsupport * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    double* prekk = (double*)malloc(sizeof(double) * (size_t)out_size * ksize);
    int* bounds = (int*)malloc(sizeof(int) * 2 * (size_t)out_size);
    int32_t* kk = (int32_t*)malloc(sizeof(int32_t) * (size_t)out_size * ksize);
    for (int xx = 0; xx < out_size; xx++) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double* k = &prekk[(size_t)xx * ksize];
        int x;
        for (x = 0; x < xmax; x++) {
            const double w = filt((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        bounds[xx * 2 + 0] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    for (size_t i = 0; i < (size_t)out_size * ksize; i++) {
        if (prekk[i] < 0) kk[i] = (int32_t)(-0.5 + prekk[i] * (1 << PRECISION_BITS));
        else kk[i] = (int32_t)(0.5 + prekk[i] * (1 << PRECISION_BITS));
    }
    free(prekk);
    *bounds_out = bounds;
    *kk_out = kk;
    return ksize;
}

void orc_free(void* p) { free(p); }

/* Full PIL.Image.resize((out_w, out_h), resample=filter) of an interleaved uint8 image with `ch`
 * channels (1..4); src rows are `src_stride` bytes apart; dst is dense [out_h, out_w, ch].
 * Mirrors ImagingResample: a pass is skipped when that axis keeps its size (and the box is the
 * whole image), both skipped -> plain copy. */
int orc_resize_u8(const uint8_t* src, int in_h, int in_w, int ch, long src_stride, int out_h, int out_w, int filter, uint8_t* dst) {
    if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || ch < 1 || ch > 4) return -1;
    const int need_h = out_w != in_w, need_v = out_h != in_h;
    const uint8_t* cur = src;
    long cur_stride = src_stride;
    uint8_t* tmp = NULL;
    if (need_h) {
        int* bounds; int32_t* kk;
        const int ksize = orc_precompute_coeffs(in_w, 0.0f, (float)in_w, out_w, filter, &bounds, &kk);
        uint8_t* out = need_v ? (tmp = (uint8_t*)malloc((size_t)in_h * out_w * ch)) : dst;
        for (int y = 0; y < in_h; y++)
            for (int xx = 0; xx < out_w; xx++) {
                const int xmin = bounds[xx * 2], xmax = bounds[xx * 2 + 1];
                const int32_t* k = &kk[(size_t)xx * ksize];
                for (int c = 0; c < ch; c++) {
                    int ss = 1 << (PRECISION_BITS - 1);
                    for (int x = 0; x < xmax; x++) ss += (int)cur[(size_t)y * cur_stride + (size_t)(x + xmin) * ch + c] * k[x];
                    out[((size_t)y * out_w + xx) * ch + c] = clip8(ss);
                }
            }
        free(bounds); free(kk);
        cur = out; cur_stride = (long)out_w * ch;
    }
    if (need_v) {
        int* bounds; int32_t* kk;
        const int ksize = orc_precompute_coeffs(in_h, 0.0f, (float)in_h, out_h, filter, &bounds, &kk);
        for (int yy = 0; yy < out_h; yy++) {
            const int ymin = bounds[yy * 2], ymax = bounds[yy * 2 + 1];
            const int32_t* k = &kk[(size_t)yy * ksize];
            for (int x = 0; x < out_w * ch; x++) {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int y = 0; y < ymax; y++) ss += (int)cur[(size_t)(y + ymin) * cur_stride + x] * k[y];
                dst[(size_t)yy * out_w * ch + x] = clip8(ss);
            }
        }
        free(bounds); free(kk);
    }
    if (!need_h && !need_v)
        for (int y = 0; y < in_h; y++) memcpy(dst + (size_t)y * in_w * ch, src + (size_t)y * src_stride, (size_t)in_w * ch);
    free(tmp);
    return 0;
}

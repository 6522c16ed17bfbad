/*
 * TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
 *
 * CPU restatement of the greedy box-NMS that the reference reaches through
 *   multipoint/utils/utils.py:103  torchvision.ops.boxes.batched_nms(boxes, scores, idxs, iou)
 *   multipoint/utils/utils.py:106  torchvision.ops.nms(boxes, scores, iou)
 * torchvision is a third-party dependency of the reference (requirements.txt:6, unpinned) whose
 * source is NOT under /root/reference, so this file restates its published CPU algorithm
 * (torchvision/csrc/ops/cpu/nms_kernel.cpp, nms_kernel_impl<float>):
 *
 *   order = argsort(scores, descending, stable)           -- ties: lower candidate index first
 *   for _i in order: if suppressed[i] continue; keep i;
 *       for every later _j: ovr = inter / (area_i + area_j - inter);  suppress j if ovr > iou
 *   with inter = max(0, xx2-xx1) * max(0, yy2-yy1), everything in fp32 -- EXCEPT the last comparison: the kernel's signature is
 *   nms_kernel_impl(const at::Tensor& dets, const at::Tensor& scores, double iou_threshold), so `ovr > iou_threshold` promotes
 *   the fp32 `ovr` to double and compares it with the caller's Python float (0.1 is the double 0.1, not 0.1f).  The two
 *   comparisons differ exactly when ovr == (float)iou, e.g. size 11, iou 0.1, offset (9, 0): inter / union = 22 / 220 rounds to
 *   0.1f > 0.1 -- torchvision suppresses (tests/test_oracle_nms.py::test_threshold_is_compared_in_double).
 *
 * PARITY UNPINNED at this third-party boundary: the reference holds no test/golden vector for
 * box_nms; the tie-break rule stated in DESIGN.md is
 * (score descending, candidate index ascending) = row-major flat index ascending.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libnms_greedy.so oracle/nms_greedy.c
 * (-O2 without -ffast-math keeps strict IEEE fp32 semantics: no contraction on x86-64 by default)
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { float s; int64_t i; } item_t;

static int cmp_desc_stable(const void *a, const void *b)
{
    const item_t *x = (const item_t *)a, *y = (const item_t *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);      /* stable: original index ascending */
}

/* boxes: n x 4 as (c0_lo, c1_lo, c0_hi, c1_hi); returns number kept; keep[] in visiting order
 * (descending score) exactly like torchvision.ops.nms. */
int64_t oracle_nms_greedy(const float *boxes, const float *scores, int64_t n, double iou,
                          int64_t *keep)
{
    if (n <= 0) return 0;
    item_t *order = (item_t *)malloc(sizeof(item_t) * (size_t)n);
    uint8_t *sup = (uint8_t *)calloc((size_t)n, 1);
    float *area = (float *)malloc(sizeof(float) * (size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        order[k].s = scores[k];
        order[k].i = k;
        const float *b = boxes + 4 * k;
        area[k] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    qsort(order, (size_t)n, sizeof(item_t), cmp_desc_stable);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        const int64_t i = order[_i].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1];
        const float ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float iarea = area[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j].i;
            if (sup[j]) continue;
            const float *b = boxes + 4 * j;
            const float xx1 = ix1 > b[0] ? ix1 : b[0];
            const float yy1 = iy1 > b[1] ? iy1 : b[1];
            const float xx2 = ix2 < b[2] ? ix2 : b[2];
            const float yy2 = iy2 < b[3] ? iy2 : b[3];
            float w = xx2 - xx1; if (w < 0.f) w = 0.f;
            float h = yy2 - yy1; if (h < 0.f) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (iarea + area[j] - inter);
            if ((double)ovr > iou) sup[j] = 1;      /* fp32 ovr against the DOUBLE threshold, as torchvision */
        }
    }
    free(order); free(sup); free(area);
    return nk;
}

/*
 * A host program in plain C over the C ABI of libmultipoint_hip.so -- no Python, no PyTorch: what a non-Python caller
 * of the reference's hot path (predict_align_image_pair.py:111-190: forward x2 -> box_nms -> nonzero ->
 * interpolate_descriptors -> get_matches) links against.  Device memory comes from the HIP runtime directly.
 *
 *   c_host_demo <weights.bin> <images.bin> <out.bin> [topk]
 *
 *   weights.bin  int32 n; n x { int32 name_len; char name[name_len]; int64 numel; float data[numel] }
 *                (the reference state_dict, torch.save(net.state_dict()), train.py:161-173, as flat fp32)
 *   images.bin   int32 B, H, W; float images[B][H][W]   (interleaved pairs: 2p = optical, 2p+1 = thermal)
 *   out.bin      int32 B, H, W, D, K; float prob[B][H][W]; float desc[B][H/8][W/8][D]; int32 kp_count[B];
 *                int32 kp_yx[B][K][2]; float kp_desc[B][K][D]; int32 match_idx[B/2][K]; int32 match_count[B/2]
 *
 * Build:  gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host_demo.c \
 *             -Lmultipoint_amd -lmultipoint_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/multipoint_amd -o c_host_demo
 * (tests/test_gpu_c_abi.py builds and runs it on the GPU box and compares with the Python host path bit for bit.)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "multipoint_hip.h"

#define CHECK_MP(call)                                                                             \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != MP_OK) {                                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mp_last_error(handle));            \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)
#define CHECK_HIP(call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                      \
            return 3;                                                                              \
        }                                                                                          \
    } while (0)

static int read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv)
{
    mp_handle* handle = NULL;
    if (argc < 4) {
        fprintf(stderr, "usage: %s weights.bin images.bin out.bin [topk]\n", argv[0]);
        return 1;
    }
    const int topk = argc > 4 ? atoi(argv[4]) : 300;

    /* ---- state_dict ---- */
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int n = 0;
    if (read_exact(f, &n, 4) || n <= 0 || n > 4096) { fprintf(stderr, "bad weights file\n"); return 1; }
    mp_tensor* tensors = (mp_tensor*)calloc((size_t)n, sizeof(mp_tensor));
    for (int i = 0; i < n; ++i) {
        int len = 0;
        long long numel = 0;
        if (read_exact(f, &len, 4) || len <= 0 || len > 255) { fprintf(stderr, "bad tensor name\n"); return 1; }
        char* name = (char*)calloc((size_t)len + 1, 1);
        if (read_exact(f, name, (size_t)len) || read_exact(f, &numel, 8) || numel < 0) return 1;
        float* data = (float*)malloc((size_t)(numel > 0 ? numel : 1) * sizeof(float));
        if (read_exact(f, data, (size_t)numel * sizeof(float))) { fprintf(stderr, "short tensor %s\n", name); return 1; }
        tensors[i].name = name; tensors[i].data = data; tensors[i].numel = numel;
    }
    fclose(f);

    /* ---- images ---- */
    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    int dims[3];
    if (read_exact(f, dims, 12)) return 1;
    const int B = dims[0], H = dims[1], W = dims[2];
    if (B <= 0 || (B & 1) || H % 8 || W % 8) { fprintf(stderr, "need an even number of images, H and W multiples of 8\n"); return 1; }
    const size_t npx = (size_t)B * H * W;
    float* images = (float*)malloc(npx * sizeof(float));
    if (read_exact(f, images, npx * sizeof(float))) return 1;
    fclose(f);

    /* ---- model: the shipped model_weights/multipoint/params.yaml ---- */
    mp_model_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.descriptor_head = 1; cfg.descriptor_size = 64; cfg.normalize_descriptors = 1; cfg.final_batchnorm = 1;
    cfg.reflection_pad = 1; cfg.double_convolution = 1; cfg.batchnorm = 1;
    CHECK_MP(mp_create(&handle, 0));
    printf("%s\n", mp_version());
    CHECK_MP(mp_load_weights(handle, &cfg, tensors, n));

    const int D = cfg.descriptor_size, Hc = H / 8, Wc = W / 8, K = topk, P = B / 2;
    float *d_img, *d_prob, *d_desc, *d_kpdesc, *d_mdist;
    int *d_kp, *d_cnt, *d_midx, *d_mcnt;
    const size_t ndesc = (size_t)B * Hc * Wc * D;
    CHECK_HIP(hipMalloc((void**)&d_img, npx * 4));
    CHECK_HIP(hipMalloc((void**)&d_prob, npx * 4));
    CHECK_HIP(hipMalloc((void**)&d_desc, ndesc * 4));
    CHECK_HIP(hipMalloc((void**)&d_kp, (size_t)B * K * 2 * 4));
    CHECK_HIP(hipMalloc((void**)&d_cnt, (size_t)B * 4));
    CHECK_HIP(hipMalloc((void**)&d_kpdesc, (size_t)B * K * D * 4));
    CHECK_HIP(hipMalloc((void**)&d_midx, (size_t)P * K * 4));
    CHECK_HIP(hipMalloc((void**)&d_mdist, (size_t)P * K * 4));
    CHECK_HIP(hipMalloc((void**)&d_mcnt, (size_t)P * 4));
    CHECK_HIP(hipMemset(d_kp, 0, (size_t)B * K * 2 * 4));
    CHECK_HIP(hipMemset(d_kpdesc, 0, (size_t)B * K * D * 4));
    CHECK_HIP(hipMemset(d_midx, 0xff, (size_t)P * K * 4));
    CHECK_HIP(hipMemcpy(d_img, images, npx * 4, hipMemcpyHostToDevice));

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    /* MultiPoint.forward (MultiPoint.py:99-135) */
    CHECK_MP(mp_forward(handle, d_img, NULL, B, H, W, d_prob, NULL, d_desc, stream));
    /* box_nms(size 4, thr 0.015, iou 0.1, top-k) + torch.nonzero (utils.py:78-122, predict_align_image_pair.py:170) */
    CHECK_MP(mp_detect_keypoints(handle, d_prob, NULL, B, H, W, 4.f, 0.015f, 0.1, topk, K, d_kp, NULL, d_cnt, 0, stream));
    /* interpolate_descriptors (utils.py:159-167) */
    CHECK_MP(mp_sample_descriptors(handle, d_desc, B, Hc, Wc, D, H, W, d_kp, d_cnt, K, d_kpdesc, stream));
    /* get_matches 'bfmatcher' crossCheck (matching.py:4-33): pair p = images 2p, 2p+1 */
    CHECK_MP(mp_match_mutual_nn(handle, d_kpdesc, d_cnt, d_kpdesc + (size_t)K * D, d_cnt + 1, 2LL * K * D, 2, P, K, D, -1.f,
                                d_midx, d_mdist, d_mcnt, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    /* ---- results ---- */
    float* prob = (float*)malloc(npx * 4);
    float* desc = (float*)malloc(ndesc * 4);
    float* kpdesc = (float*)malloc((size_t)B * K * D * 4);
    int* kp = (int*)malloc((size_t)B * K * 2 * 4);
    int* cnt = (int*)malloc((size_t)B * 4);
    int* midx = (int*)malloc((size_t)P * K * 4);
    int* mcnt = (int*)malloc((size_t)P * 4);
    CHECK_HIP(hipMemcpy(prob, d_prob, npx * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(desc, d_desc, ndesc * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(kpdesc, d_kpdesc, (size_t)B * K * D * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(kp, d_kp, (size_t)B * K * 2 * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(cnt, d_cnt, (size_t)B * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(midx, d_midx, (size_t)P * K * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(mcnt, d_mcnt, (size_t)P * 4, hipMemcpyDeviceToHost));
    for (int p = 0; p < P; ++p)
        printf("pair %d: %d optical keypoints, %d thermal keypoints, %d mutual matches\n", p, cnt[2 * p], cnt[2 * p + 1], mcnt[p]);
    f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 1; }
    const int hdr[5] = {B, H, W, D, K};
    fwrite(hdr, 4, 5, f);
    fwrite(prob, 4, npx, f);
    fwrite(desc, 4, ndesc, f);
    fwrite(cnt, 4, (size_t)B, f);
    fwrite(kp, 4, (size_t)B * K * 2, f);
    fwrite(kpdesc, 4, (size_t)B * K * D, f);
    fwrite(midx, 4, (size_t)P * K, f);
    fwrite(mcnt, 4, (size_t)P, f);
    fclose(f);

    /* wrong use fails loudly with a message */
    if (mp_forward(handle, d_img, NULL, B, H + 4, W, d_prob, NULL, d_desc, stream) == MP_OK) {
        fprintf(stderr, "expected an error for H not divisible by 8\n");
        return 4;
    }
    printf("error path: %s\n", mp_last_error(handle));
    mp_destroy(handle);
    return 0;
}

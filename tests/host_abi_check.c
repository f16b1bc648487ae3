/* host_abi_check.c -- HOST-ONLY driver of the C ABI (include/detr_hip.h), built against the AddressSanitizer build of the
 * library (`make -C detr-tensorflow_amd asan-check`).  It exercises, without a GPU, everything the library does on the host
 * before a launch: the layout self-check, the scratch-size queries and the validation / rejection paths of the descriptor
 * entry points (each must return a negative code and leave a message, never touch memory it was not given).  A stack or heap
 * error in that code aborts the program (ASan); the exit status is the number of failed expectations.
 * This is test infrastructure: plain C on purpose -- it is also the smallest example of a foreign binder. */
#include <stdio.h>
#include <string.h>
#include "detr_hip.h"

static int fails = 0;
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) { ++fails; printf("FAIL %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, detr_hip_last_error()); } \
    } while (0)

int main(void) {
    EXPECT(detr_hip_abi_version() == DETR_HIP_ABI_VERSION);
    EXPECT(detr_hip_reload_tuning() == 0);

    /* layout self-check: sizeof must agree with this translation unit's view of the header */
    int32_t buf[96];
    const size_t sizes[9] = {sizeof(detr_reduce_desc), sizeof(detr_gemm_desc), sizeof(detr_conv3x3_desc), sizeof(detr_stem_desc),
                             sizeof(detr_layernorm_desc), sizeof(detr_attn_desc), sizeof(detr_setloss_desc), sizeof(detr_input_desc),
                             sizeof(detr_postprocess_desc)};
    for (int w = 0; w < 9; ++w) {
        const int n = detr_hip_struct_layout(w, buf, 96);
        EXPECT(n > 1 && n <= 96);
        EXPECT((size_t)buf[0] == sizes[w]);
        int32_t tiny[2] = {-1, -1};                     /* a short buffer is never overrun; the count is still reported */
        EXPECT(detr_hip_struct_layout(w, tiny, 2) == n && tiny[0] == buf[0] && tiny[1] == buf[1]);
    }
    EXPECT(detr_hip_struct_layout(-1, buf, 96) < 0 && strlen(detr_hip_last_error()) > 0);

    /* scratch sizing */
    detr_gemm_desc g;
    memset(&g, 0, sizeof g);
    EXPECT(detr_hip_workspace_bytes_gemm(NULL) < 0);
    EXPECT(detr_hip_workspace_bytes_gemm(&g) < 0);                       /* zero shape */
    g.M = 64; g.N = 256; g.K = 534400; g.split_k = 256; g.compute = 1; g.batch = 1;
    EXPECT(detr_hip_workspace_bytes_gemm(&g) == (int64_t)254 * 64 * 256 * 4);     /* 16700 K tiles, 66 per split: 254 non-empty splits */
    g.batch = 2;
    EXPECT(detr_hip_workspace_bytes_gemm(&g) == 0);
    detr_conv3x3_desc c;
    memset(&c, 0, sizeof c);
    EXPECT(detr_hip_workspace_bytes_conv3x3(&c, 2) < 0);
    c.N = 8; c.Hi = c.Ho = 200; c.Wi = c.Wo = 334; c.Ci = c.Co = 64; c.stride = 1; c.pad = 1; c.compute = 1;
    EXPECT(detr_hip_workspace_bytes_conv3x3(&c, 0) == 0);
    EXPECT(detr_hip_workspace_bytes_conv3x3(&c, 2) == (int64_t)503 * 9 * 64 * 64 * 4);   /* 17600 pixel units, 35 per workgroup */
    c.compute = 0;
    EXPECT(detr_hip_workspace_bytes_conv3x3(&c, 2) > 0);
    detr_stem_desc s;
    memset(&s, 0, sizeof s);
    EXPECT(detr_hip_workspace_bytes_stem(&s, 2) < 0);
    s.N = 2; s.H = 96; s.W = 128; s.Ho = 48; s.Wo = 64; s.split = 4;
    EXPECT(detr_hip_workspace_bytes_stem(&s, 2) == (int64_t)4 * 147 * 64 * 4);
    detr_layernorm_desc ln;
    memset(&ln, 0, sizeof ln);
    EXPECT(detr_hip_workspace_bytes_layernorm(&ln) < 0);
    ln.rows = 800; ln.C = 256;
    EXPECT(detr_hip_workspace_bytes_layernorm(&ln) == (int64_t)100 * 2 * 256 * 4);
    {   /* tile plan of the ring GEMM: M = 33600 x N = 256 as 128-column panels, two workgroups per CU (<= 80 KB of LDS each) */
        int32_t rp[8];
        EXPECT(detr_hip_gemm_ring_plan(33600, 256, 1024, rp) == 1);
        EXPECT(rp[1] == 1 && rp[5] == 2 && rp[6] == rp[4] * rp[5] && rp[3] * rp[4] >= 33600 && rp[3] <= 64 * rp[0] && rp[7] <= 80 * 1024);
        EXPECT(detr_hip_gemm_ring_plan(33600, 256, 1000, rp) == 0);        /* K % 64 != 0: no plan */
        EXPECT(detr_hip_gemm_ring_plan(8400, 64, 512, NULL) == 0);         /* narrower than one column panel */
    }

    /* rejection paths: every descriptor entry point validates before it launches (no GPU is touched here) */
    EXPECT(detr_hip_gemm_f32(NULL, NULL) < 0);
    memset(&g, 0, sizeof g);
    EXPECT(detr_hip_gemm_f32(&g, NULL) < 0);                              /* bad shape */
    g.M = g.N = g.K = 64;
    EXPECT(detr_hip_gemm_f32(&g, NULL) < 0);                              /* null operands */
    EXPECT(detr_hip_gemm_group_f32(NULL, 3, NULL) < 0);
    EXPECT(detr_hip_gemm_group_f32(&g, 0, NULL) < 0);
    EXPECT(detr_hip_splitk_reduce_many(NULL, 1, NULL) < 0);
    EXPECT(detr_hip_conv3x3_f32(NULL, 0, NULL) < 0);
    memset(&c, 0, sizeof c);
    c.stride = 1;
    EXPECT(detr_hip_conv3x3_f32(&c, 7, NULL) < 0);                        /* bad mode */
    c.stride = 3;
    EXPECT(detr_hip_conv3x3_f32(&c, 0, NULL) < 0);
    EXPECT(detr_hip_stem_conv7x7_f32(NULL, 0, NULL) < 0);
    memset(&s, 0, sizeof s);
    EXPECT(detr_hip_stem_conv7x7_f32(&s, 1, NULL) < 0);                   /* mode 1 does not exist */
    EXPECT(detr_hip_stem_conv7x7_f32(&s, 0, NULL) < 0);                   /* null operands */
    detr_attn_desc a;
    memset(&a, 0, sizeof a);
    EXPECT(detr_hip_attention_fwd(&a, NULL) < 0);
    EXPECT(detr_hip_attention_bwd(&a, NULL) < 0);
    memset(&ln, 0, sizeof ln);
    EXPECT(detr_hip_layernorm_fwd(&ln, NULL) < 0);
    EXPECT(detr_hip_layernorm_bwd(&ln, NULL) < 0);
    detr_postprocess_desc pp;
    memset(&pp, 0, sizeof pp);
    EXPECT(detr_hip_postprocess(&pp, NULL) < 0);
    detr_input_desc in;
    memset(&in, 0, sizeof in);
    EXPECT(detr_hip_input_stage(&in, NULL) < 0);
    EXPECT(strlen(detr_hip_last_error()) > 0);
    printf("host_abi_check: %d failed expectation(s)\n", fails);
    return fails;
}

/* A plain-C consumer of include/nerf_mi355.h (no Python, no torch): renders a 16x16 image with seeded
 * Glorot-like weights through the host-memory entry point and prints a checksum + a few pixels.
 * Built and run by tests/test_gpu_parity.py::test_c_abi_client on the GPU box:
 *   gcc -O2 -Iinclude tests/abi_c_client.c -o <tmp>/abi_c_client -L<lib dir> -lnerf_mi355 -Wl,-rpath,<lib dir> -lm
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "nerf_mi355.h"

int main(int argc, char** argv) {
    /* weights + pose come from a file written by the test (raw fp32): blob_c | blob_f | c2w(16) */
    if (argc < 2) { fprintf(stderr, "usage: %s weights.bin\n", argv[0]); return 2; }
    nerf_config cfg = {5, 4, 2, 256, 128, 0.05f, 2.0f / 3.0f, 5.0f / 3.0f, NERF_PRECISION_FP32, 0};
    const size_t nb = nerf_blob_size(&cfg);
    float* buf = (float*)malloc((2 * nb + 16) * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(buf, sizeof(float), 2 * nb + 16, f) != 2 * nb + 16) { fprintf(stderr, "bad weights file\n"); return 2; }
    fclose(f);
    nerf_ctx* ctx = NULL;
    if (nerf_ctx_create(&cfg, &ctx)) { fprintf(stderr, "create: %s\n", nerf_last_error()); return 1; }
    if (nerf_load_weights(ctx, NERF_NET_COARSE, buf, nb) || nerf_load_weights(ctx, NERF_NET_FINE, buf + nb, nb)) {
        fprintf(stderr, "load: %s\n", nerf_last_error()); return 1;
    }
    /* a wrong blob size must be refused with a message, not crash */
    if (nerf_load_weights(ctx, NERF_NET_COARSE, buf, nb - 1) == 0) { fprintf(stderr, "size check missing\n"); return 1; }
    const int H = 16, W = 16;
    /* page-locked outputs (nerf_host_alloc): the device writes them by DMA beside the kernels */
    float *rgb = NULL, *depth = NULL;
    if (nerf_abi_version() != NERF_ABI_VERSION || nerf_host_alloc(sizeof(float) * H * W * 3, (void**)&rgb) ||
        nerf_host_alloc(sizeof(float) * H * W, (void**)&depth)) {
        fprintf(stderr, "host_alloc: %s\n", nerf_last_error()); return 1;
    }
    nerf_outputs out = {0};
    out.rgb = rgb; out.depth = depth;
    if (nerf_render_image(ctx, buf + 2 * nb, 0.6911112f, H, W, 0, 0, 0, 64, 128, NULL, NULL, 12345u, &out, NERF_MEM_HOST)) {
        fprintf(stderr, "render: %s\n", nerf_last_error()); return 1;
    }
    double sum = 0;
    for (int i = 0; i < H * W * 3; ++i) sum += rgb[i];
    printf("checksum %.9f\n", sum);
    printf("pixel0 %.9g %.9g %.9g depth0 %.9g\n", rgb[0], rgb[1], rgb[2], depth[0]);
    printf("pixel255 %.9g %.9g %.9g depth255 %.9g\n", rgb[255 * 3], rgb[255 * 3 + 1], rgb[255 * 3 + 2], depth[255]);
    /* the same frame through the library's own RCCL assembly (no torch in this process: librccl is found by dlopen).
     * Usage with peers: abi_c_client weights.bin <rank> <world> <id file> -- rank 0 writes the communicator id to the file,
     * the others wait for it.  Default: a one-rank communicator.  If the communicator cannot be created (no RCCL, a second
     * communicator refused, a peer missing) the client SAYS SO on stdout ("comm_unavailable <reason>") and falls back to the
     * single-rank render, like bench.py does -- so a first real N > 1 failure is attributable to the communicator and not
     * to the renderer. */
    {
        const int rank = argc >= 5 ? atoi(argv[2]) : 0, world = argc >= 5 ? atoi(argv[3]) : 1;
        char id[NERF_COMM_ID_BYTES];
        float* full = (float*)malloc(sizeof(float) * H * W * 3);
        float* full_depth = (float*)malloc(sizeof(float) * H * W);
        int have_comm = 0;
        char why[600] = "";
        if (rank == 0) {
            if (nerf_comm_unique_id(id)) snprintf(why, sizeof why, "nerf_comm_unique_id: %s", nerf_last_error());
            else if (argc >= 5) {
                FILE* g = fopen(argv[4], "wb");
                if (!g || fwrite(id, 1, sizeof id, g) != sizeof id) snprintf(why, sizeof why, "cannot write %s", argv[4]);
                if (g) fclose(g);
            }
        } else {
            FILE* g = NULL;
            for (int tries = 0; tries < 600 && !g; ++tries) {                 /* up to 60 s for rank 0 */
                g = fopen(argv[4], "rb");
                if (g && fread(id, 1, sizeof id, g) != sizeof id) { fclose(g); g = NULL; }
                if (!g) { struct timespec ts = {0, 100000000}; nanosleep(&ts, NULL); }
            }
            if (!g) snprintf(why, sizeof why, "rank 0 never published the communicator id in %s", argv[4]);
            else fclose(g);
        }
        if (!why[0]) {
            if (nerf_comm_init(ctx, id, rank, world)) snprintf(why, sizeof why, "nerf_comm_init: %s", nerf_last_error());
            else have_comm = 1;
        }
        nerf_outputs both = {0};
        both.rgb = full; both.depth = full_depth;
        if (have_comm) {
            float* rgb_only = (float*)malloc(sizeof(float) * H * W * 3);
            if (nerf_render_image_sharded(ctx, buf + 2 * nb, 0.6911112f, H, W, 0, 64, 128, 12345u, rgb_only, NERF_MEM_HOST) ||
                nerf_render_image_sharded_outputs(ctx, buf + 2 * nb, 0.6911112f, H, W, 0, 64, 128, 12345u, &both,
                                                  NERF_MEM_HOST)) {
                fprintf(stderr, "sharded render: %s\n", nerf_last_error()); return 1;
            }
            int same = 1;
            for (int i = 0; i < H * W * 3; ++i) same &= full[i] == rgb[i] && rgb_only[i] == rgb[i];
            for (int i = 0; i < H * W; ++i) same &= full_depth[i] == depth[i];
            printf("sharded_equal %d world %d\n", same, world);
            nerf_comm_destroy(ctx);
            free(rgb_only);
        } else {
            printf("comm_unavailable %s\n", why);
            if (nerf_render_image(ctx, buf + 2 * nb, 0.6911112f, H, W, 0, 0, 0, 64, 128, NULL, NULL, 12345u, &both,
                                  NERF_MEM_HOST)) {
                fprintf(stderr, "single-rank fallback render: %s\n", nerf_last_error()); return 1;
            }
            int same = 1;
            for (int i = 0; i < H * W * 3; ++i) same &= full[i] == rgb[i];
            printf("single_rank_fallback_equal %d\n", same);
        }
        free(full); free(full_depth);
    }
    /* three training steps (NeRF.train_step) on the rays of that frame towards a constant colour: the loss must fall */
    {
        const int N = H * W;
        float* dirs = (float*)malloc(sizeof(float) * N * 4);
        float* orig = (float*)malloc(sizeof(float) * N * 4);
        float* tgt = (float*)malloc(sizeof(float) * N * 3);
        if (nerf_get_rays_directions(ctx, buf + 2 * nb, 0.6911112f, H, W, dirs, NERF_MEM_HOST)) {
            fprintf(stderr, "dirs: %s\n", nerf_last_error()); return 1;
        }
        for (int i = 0; i < N; ++i) {
            for (int k = 0; k < 4; ++k) orig[i * 4 + k] = buf[2 * nb + k * 4 + 3];      /* c2w[:, 3] */
            tgt[i * 3] = 0.8f; tgt[i * 3 + 1] = 0.3f; tgt[i * 3 + 2] = 0.1f;
        }
        nerf_train_config tc = {5e-4f, 0.9f, 0.999f, 1e-7f, 1};
        float m0[3], m1[3];
        if (nerf_train_begin(ctx, &tc) ||
            nerf_train_step(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 1u, m0, NERF_MEM_HOST) ||
            nerf_train_step(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 1u, m1, NERF_MEM_HOST) ||
            nerf_train_step(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 1u, m1, NERF_MEM_HOST) || nerf_train_end(ctx)) {
            fprintf(stderr, "train: %s\n", nerf_last_error()); return 1;
        }
        printf("train_loss %.9g %.9g\n", m0[0], m1[0]);
        /* ABI 5: the ray loss DietNeRF's tape differentiates counts the coarse MSE twice (src/DietNeRF.py:163-171): the same
         * gradients call under (1, 1) and (2, 1) -- the loss metric moves by MSE_c = 10^(-psnr_coarse / 10), the PSNRs do not;
         * nerf_train_begin restores (1, 1) and bad weights are refused */
        float ma[3], mb[3], mc[3];
        if (nerf_train_begin(ctx, &tc) ||
            nerf_train_gradients(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 5u, NULL, NULL, ma, NERF_MEM_HOST) ||
            nerf_train_set_loss_weights(ctx, 2.0f, 1.0f) ||
            nerf_train_gradients(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 5u, NULL, NULL, mb, NERF_MEM_HOST) ||
            nerf_train_end(ctx) || nerf_train_begin(ctx, &tc) ||
            nerf_train_gradients(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 5u, NULL, NULL, mc, NERF_MEM_HOST)) {
            fprintf(stderr, "loss weights: %s\n", nerf_last_error()); return 1;
        }
        const int refused = nerf_train_set_loss_weights(ctx, -1.0f, 1.0f) != 0;
        if (nerf_train_end(ctx)) { fprintf(stderr, "loss weights: %s\n", nerf_last_error()); return 1; }
        printf("loss_weights %.9g %.9g %.9g psnr %.9g %.9g refused %d\n", ma[0], mb[0], mc[0], ma[1], mb[1], refused);
        /* the same under the reference's production policy: mixed_float16 with dynamic loss scaling; an infinite target is a
         * SKIPPED step that halves the scale (ABI 2/3) */
        nerf_train_config tm = {5e-4f, 0.9f, 0.999f, 1e-7f, 1, 1, 1024.0f, 0};
        float scale = 0.f;
        int64_t applied = -1, skipped = -1;
        if (nerf_train_begin(ctx, &tm) ||
            nerf_train_step(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 2u, m0, NERF_MEM_HOST)) {
            fprintf(stderr, "train mixed: %s\n", nerf_last_error()); return 1;
        }
        tgt[0] = INFINITY;
        if (nerf_train_step(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 3u, m1, NERF_MEM_HOST) ||
            nerf_train_loss_scale(ctx, &scale, &applied, &skipped) || nerf_train_end(ctx)) {
            fprintf(stderr, "train mixed: %s\n", nerf_last_error()); return 1;
        }
        printf("mixed %.9g %g %lld %lld\n", m0[0], scale, (long long)applied, (long long)skipped);
        /* ABI 4: DietNeRF's step under that policy from C -- ray-loss gradients, plus the backward through NeRF.render of a
         * caller-supplied dL/d(rgb) accumulated on top (both unscaled), ONE verdict in nerf_train_apply; and the epoch
         * means of the metrics read from the device-side sums */
        tgt[0] = 0.8f;
        float* d_rgb = (float*)malloc(sizeof(float) * N * 3);
        for (int i = 0; i < N * 3; ++i) d_rgb[i] = 1e-3f * (float)((i % 7) - 3);
        double sums[3];
        int64_t steps = -1;
        if (nerf_train_begin(ctx, &tm) || nerf_train_read_metric_sums(ctx, sums, &steps) ||
            nerf_train_gradients(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 4u, NULL, NULL, NULL, NERF_MEM_HOST) ||
            nerf_train_render_gradients(ctx, orig, dirs, d_rgb, N, 16, 16, NULL, NULL, 4u, 0, 1, NULL, NULL, NULL,
                                        NERF_MEM_HOST) ||
            nerf_train_apply(ctx, NULL, NULL, NERF_MEM_HOST) || nerf_train_read_metric_sums(ctx, sums, &steps) ||
            nerf_train_loss_scale(ctx, &scale, &applied, &skipped) || nerf_train_end(ctx)) {
            fprintf(stderr, "dietnerf step: %s\n", nerf_last_error()); return 1;
        }
        printf("dietnerf_mixed %g %lld %lld metric_steps %lld loss %.9g\n", scale, (long long)applied, (long long)skipped,
               (long long)steps, sums[0]);
        /* ABI 5: the same step with the image's activations KEPT between forward and backward (one forward, as under the
         * reference's tape): nerf_train_render_forward hands out the rgb the caller's embedding network sees, its d_rgb goes
         * into nerf_train_render_backward of the same slot; a second backward on the consumed slot is refused */
        float* rgb_fw = (float*)malloc(sizeof(float) * N * 3);
        if (nerf_train_begin(ctx, &tm) ||
            nerf_train_gradients(ctx, orig, dirs, tgt, N, 16, 16, NULL, NULL, 4u, NULL, NULL, NULL, NERF_MEM_HOST) ||
            nerf_train_render_forward(ctx, 0, orig, dirs, N, 16, 16, NULL, NULL, 4u, 0, rgb_fw, NERF_MEM_HOST) ||
            nerf_train_render_backward(ctx, 0, d_rgb, 1, NULL, NULL, NERF_MEM_HOST)) {
            fprintf(stderr, "dietnerf slots: %s\n", nerf_last_error()); return 1;
        }
        const int consumed = nerf_train_render_backward(ctx, 0, d_rgb, 1, NULL, NULL, NERF_MEM_HOST) != 0;
        if (nerf_train_apply(ctx, NULL, NULL, NERF_MEM_HOST) || nerf_train_loss_scale(ctx, &scale, &applied, &skipped) ||
            nerf_train_render_release(ctx) || nerf_train_end(ctx)) {
            fprintf(stderr, "dietnerf slots: %s\n", nerf_last_error()); return 1;
        }
        int finite_rgb = 1;
        for (int i = 0; i < N * 3; ++i) finite_rgb &= rgb_fw[i] >= 0.f && rgb_fw[i] <= 1.f;
        printf("dietnerf_slots %g %lld %lld consumed_refused %d rgb_in_range %d\n", scale, (long long)applied,
               (long long)skipped, consumed, finite_rgb);
        free(rgb_fw);
        free(d_rgb);
        free(dirs); free(orig); free(tgt);
    }
    nerf_ctx_destroy(ctx);
    nerf_host_free(rgb); nerf_host_free(depth);
    free(buf);
    return 0;
}

/* A host written against include/rtmi.h alone, in C: builds the reference's "debug" scene
 * (crates/rt/src/scene/examples/debug.rs:6-16: two diffuse spheres), asks for one closest hit, renders a small frame
 * on rank 0 of a one-rank job and prints what it got.  Build: cc -std=c99 -Iinclude examples/abi_smoke.c
 * -Lraytracing_amd -lrtmi -lm -o abi_smoke; run with LD_LIBRARY_PATH=raytracing_amd.  Exit status 0 = every call
 * succeeded and the numbers are sane. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtmi.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != RTMI_OK) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rtmi_last_error());        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    rtmi_device *dev = NULL;
    rtmi_scene *scene = NULL;
    CHECK(rtmi_device_create(0, &dev));
    CHECK(rtmi_scene_create(dev, &scene));
    rtmi_material yellow;
    memset(&yellow, 0, sizeof yellow);
    yellow.kind = RTMI_MAT_DIFFUSE;
    yellow.rgb[0] = 1.0f;
    yellow.rgb[1] = 1.0f;
    uint32_t mat = 0, g0 = 0, g1 = 0;
    CHECK(rtmi_scene_insert_material(scene, &yellow, &mat));
    const float c0[3] = {0.0f, 0.0f, -1.0f}, c1[3] = {0.5f, 0.0f, -1.0f};
    CHECK(rtmi_scene_insert_sphere(scene, mat, c0, 0.3f, &g0));
    CHECK(rtmi_scene_insert_sphere(scene, mat, c1, 0.3f, &g1));
    CHECK(rtmi_scene_commit(scene, NULL, NULL));

    rtmi_ray ray = {{0.0f, 0.0f, 0.0f}, 1e-5f, {0.0f, 0.0f, -1.0f}, INFINITY};
    rtmi_hit hit;
    CHECK(rtmi_intersect(scene, &ray, &hit, 1));
    printf("hit: geom %u t %.6f n (%.3f %.3f %.3f) material %u\n", hit.geom_id, hit.t, hit.n[0], hit.n[1], hit.n[2],
           hit.mat_id);
    if (hit.geom_id != g0 || fabsf(hit.t - 0.7f) > 1e-5f || hit.mat_id != mat) return 2;

    const uint32_t W = 64, H = 36, spp = 4;
    rtmi_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.width = W;
    cam.height = H;
    cam.focal_length = 1.0f;
    cam.half_height = tanf(0.5f * 70.0f * 3.14159265f / 180.0f); /* utils/mod.rs:26-45 */
    cam.half_width = cam.half_height * (float)W / (float)H;
    cam.rot[3] = 1.0f;
    rtmi_render_params prm;
    memset(&prm, 0, sizeof prm);
    prm.spp = spp;
    prm.sample_end = spp;
    prm.max_depth = 64;

    uint32_t first = 0, stride = 0, count = 0;
    CHECK(rtmi_tile_partition(rtmi_tile_count(W, H, 32), 1, 0, &first, &stride, &count));
    rtmi_film *film = NULL, *frame = NULL;
    CHECK(rtmi_film_create_tiles(dev, 0, 0, W, H, 32, first, stride, 8, 8, &film));
    CHECK(rtmi_film_create_frame(dev, 0, 0, W, H, &frame));
    rtmi_render_stats st;
    CHECK(rtmi_render(scene, film, &cam, &prm, &st));
    CHECK(rtmi_film_gather(film, NULL, 0, frame));
    float *img = (float *)malloc(sizeof(float) * RTMI_RESOLVE_PLANES * W * H);
    if (!img) return 3;
    CHECK(rtmi_film_resolve(frame, img, 1));
    /* Color planes 9..11; the top-left pixel sees the sky: sRGB(0.5, 0.3, 1.0) */
    const float *r = img + 9 * W * H, *g = img + 10 * W * H, *b = img + 11 * W * H;
    double sum = 0.0;
    for (uint32_t i = 0; i < W * H; ++i) sum += r[i] + g[i] + b[i];
    printf("rendered %llu paths, %llu rays; corner colour %.4f %.4f %.4f; checksum %.6f\n",
           (unsigned long long)st.paths, (unsigned long long)st.rays, r[0], g[0], b[0], sum);
    const int ok = st.paths == (unsigned long long)W * H * spp && fabsf(r[0] - 0.7354f) < 2e-3f &&
                   fabsf(g[0] - 0.5838f) < 2e-3f && fabsf(b[0] - 1.0f) < 1e-6f;
    free(img);
    rtmi_film_destroy(film);
    rtmi_film_destroy(frame);
    rtmi_scene_destroy(scene);
    rtmi_device_destroy(dev);
    return ok ? 0 : 4;
}

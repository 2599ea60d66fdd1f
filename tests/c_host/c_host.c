/* A host written in plain C99 against include/xrt.h alone (no Python, no C++): what any FFI -- P/Invoke from the reference's C#
 * (csharp/XrtNative.cs), cgo, JNI -- does.  Loads a scene file, builds it, renders one frame into a host Color[] and writes it.
 *   gcc -std=c99 -I include tests/c_host/c_host.c -L xna-ray-trace_amd/csrc -lxrt -Wl,-rpath,... -o c_host
 *   c_host <scene.xrts> <frame parameters> <out.rgba>
 * frame parameters: xrt_camera, int32 n_lights, xrt_light[n_lights], xrt_render_opts -- the bytes of the C structs. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xrt.h"

#define C_HOST_MAX_LIGHTS 64   /* this host's own array; the library takes any number (RT:534-542 iterates a list) */

static int fail(const char *what, int rc) {
    fprintf(stderr, "c_host: %s failed (%d): %s\n", what, rc, xrt_last_error());
    return 1;
}

int main(int argc, char **argv) {
    xrt_scene *scene = NULL;
    xrt_camera cam;
    xrt_light lights[C_HOST_MAX_LIGHTS];
    xrt_render_opts opts;
    xrt_stats stats;
    int32_t n_lights = 0;
    uint32_t *frame = NULL;
    size_t px;
    FILE *f;
    int rc, n_dev = 0;

    if (argc != 4) { fprintf(stderr, "usage: c_host <scene.xrts> <frame parameters> <out.rgba>\n"); return 2; }
    if (xrt_version() != XRT_VERSION) { fprintf(stderr, "c_host: header %d, library %d\n", XRT_VERSION, xrt_version()); return 1; }
    if ((rc = xrt_device_count(&n_dev)) != XRT_OK || n_dev < 1) return fail("xrt_device_count", rc);
    f = fopen(argv[2], "rb");
    if (!f || fread(&cam, sizeof(cam), 1, f) != 1 || fread(&n_lights, sizeof(n_lights), 1, f) != 1 || n_lights < 0 || n_lights > C_HOST_MAX_LIGHTS ||
        (n_lights && fread(lights, sizeof(xrt_light), (size_t)n_lights, f) != (size_t)n_lights) || fread(&opts, sizeof(opts), 1, f) != 1) {
        fprintf(stderr, "c_host: cannot read %s\n", argv[2]);
        return 1;
    }
    fclose(f);
    if ((rc = xrt_scene_load(0, argv[1], &scene)) != XRT_OK) return fail("xrt_scene_load", rc);
    if ((rc = xrt_scene_build(scene, 0, 0)) != XRT_OK) return fail("xrt_scene_build", rc);
    px = (size_t)cam.vp_width * (size_t)cam.vp_height;
    frame = (uint32_t *)calloc(px ? px : 1, sizeof(uint32_t));
    if (!frame) return 1;
    /* an argument error is reported, not swallowed: a NULL camera */
    if (xrt_render(scene, NULL, lights, n_lights, &opts, frame, NULL, NULL) != XRT_E_INVALID_ARG) { fprintf(stderr, "c_host: NULL camera accepted\n"); return 1; }
    memset(&stats, 0, sizeof(stats));
    if ((rc = xrt_render(scene, &cam, lights, n_lights, &opts, frame, NULL, &stats)) != XRT_OK) return fail("xrt_render", rc);
    f = fopen(argv[3], "wb");
    if (!f || fwrite(frame, sizeof(uint32_t), px, f) != px) { fprintf(stderr, "c_host: cannot write %s\n", argv[3]); return 1; }
    fclose(f);
    printf("c_host: %d x %d, %llu closest-hit + %llu shadow rays, %.3f ms on the GPU\n", (int)cam.vp_width, (int)cam.vp_height,
           (unsigned long long)stats.rays_closest, (unsigned long long)stats.rays_shadow, stats.ms_total);
    free(frame);
    if ((rc = xrt_scene_destroy(scene)) != XRT_OK) return fail("xrt_scene_destroy", rc);
    return 0;
}

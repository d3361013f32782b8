/* c_abi_check.c -- compiled as C99 by tests/test_host_cpp_api.py: include/crt_hip.h must be usable from plain C (what a
 * cgo / JNI / N-API binding sees), and the host-side entry points must work without a GPU. */
#include <stdio.h>
#include <string.h>

#include "../include/crt_hip.h"

int main(void) {
    crt_camera_params p;
    crt_camera cam;
    memset(&p, 0, sizeof p);
    p.aspect_ratio = 16.0f / 9.0f; p.image_width = 400; p.samples_per_pixel = 10; p.vfov = 20.0f;
    p.lookfrom[0] = 13; p.lookfrom[1] = 2; p.lookfrom[2] = 3; p.vup[1] = 1; p.defocus_angle = 0.6f; p.focus_dist = 10.0f;
    if (crt_camera_initialize(&p, &cam) != CRT_OK || cam.image_height != 225) return 1;
    if (crt_bvh_num_nodes(5) != 15 || crt_bvh_num_nodes(486) != 1023) return 2;
    {
        float center[15], motion[15], radius[5], albedo[15], param[5];
        int32_t mat_id[5], mat_type[5], n = 0;
        crt_bvh_node nodes[15];
        if (crt_scene_small(center, motion, radius, mat_id, mat_type, albedo, param, &n) != CRT_OK || n != 5) return 3;
        if (crt_bvh_build_host(n, center, motion, radius, 17u, nodes) != CRT_OK) return 4;
        if (nodes[0].left_child_loc != 1 || nodes[0].right_child_loc != 2) return 5;
    }
    {   /* no GPU in the test container: creation must fail with a message, not crash; on a GPU box it succeeds */
        crt_context* ctx = NULL;
        crt_group* grp = NULL;
        int32_t dev[2] = {0, 0};
        int rc = crt_create(0, 1, &ctx);
        if (rc == CRT_OK) crt_destroy(ctx);
        else if (strlen(crt_last_error(NULL)) == 0) return 6;
        rc = crt_group_create(dev, 2, 1, &grp);
        if (rc == CRT_OK) { if (crt_group_size(grp) != 2) return 7; crt_group_destroy(grp); }
        else if (strlen(crt_group_last_error(NULL)) == 0) return 8;
    }
    if (strlen(crt_build_id()) == 0 || CRT_STATIC_MAX_PIXELS != (1 << 21)) return 9;
    printf("OK\n");
    return 0;
}

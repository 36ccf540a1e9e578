/* A plain C99 caller of the C ABI (what a cgo / FFI binding sees): host-only entry points, so it also runs without a GPU.
 * With a device argument it detects the markers of a PGM file. Built and run by tests/test_cabi_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arucohip.h"

static unsigned char* read_pgm(const char* path, int* w, int* h) {
    FILE* f = fopen(path, "rb");
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    unsigned char* px;
    if (!f) return NULL;
    if (fscanf(f, "%2s %d %d %d", magic, w, h, &maxv) != 4 || strcmp(magic, "P5") != 0 || maxv != 255) {
        fclose(f);
        return NULL;
    }
    fgetc(f);
    px = (unsigned char*)malloc((size_t)*w * *h);
    if (px && fread(px, 1, (size_t)*w * *h, f) != (size_t)*w * *h) {
        free(px);
        px = NULL;
    }
    fclose(f);
    return px;
}

int main(int argc, char** argv) {
    arucohip_params_t p;
    arucohip_limits_t lim;
    double mv[16], rv[3] = {0.1, -0.2, 0.3}, tv[3] = {0.01, 0.02, 0.5};
    arucohip_default_params(&p);
    arucohip_default_limits(&lim, 1920, 1080, 4);
    if (arucohip_gl_modelview(rv, tv, mv) != ARUCOHIP_OK) return 2;
    printf("version %d thres %d/%g/%g corner %d warp %d limits %dx%dx%d marker_bytes %d mv15 %g\n", arucohip_version(), (int)p.thres_method, p.thres_param1,
           p.thres_param2, (int)p.corner_method, (int)p.warp_size, (int)lim.max_width, (int)lim.max_height, (int)lim.max_batch, (int)sizeof(arucohip_marker_t), mv[15]);
    if (argc >= 2) { /* needs a GPU */
        int w = 0, h = 0, n = 0, i, rc;
        arucohip_handle* hd = NULL;
        arucohip_marker_t out[64];
        unsigned char* px = read_pgm(argv[1], &w, &h);
        if (!px) return 3;
        rc = arucohip_create(NULL, 0, w, h, 1, &hd);
        if (rc != ARUCOHIP_OK) return 4;
        rc = arucohip_detect(hd, px, w, h, (size_t)w, NULL, NULL, 0, -1.0f, 0, out, 64, &n);
        if (rc != ARUCOHIP_OK) {
            fprintf(stderr, "%s\n", arucohip_last_error_string(hd));
            return 5;
        }
        printf("markers %d:", n);
        for (i = 0; i < n; i++) printf(" %d", (int)out[i].id);
        printf("\n");
        arucohip_destroy(hd);
        free(px);
    }
    return 0;
}

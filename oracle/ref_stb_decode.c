/*
 * ref_stb_decode.c — TEST INFRASTRUCTURE.  Harness around the reference's own vendored decoder:
 * it #includes /root/reference/include/stb_image.h where it lies (never copied into this repo),
 * calls stbi_loadf(path, &w, &h, &c, 4) exactly as the reference's load_texture_cpu does
 * (src/main.cu:52-60) and writes the RGB floats as a PFM file (rows bottom to top) that the host
 * mirror's texture loader reads.  Built only where /root/reference exists (oracle/Makefile
 * target `_ref`), output binary under oracle/_ref/ (git-ignored).
 */
#define STB_IMAGE_IMPLEMENTATION
#include REF_STB_IMAGE_H
#include <stdio.h>

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.jpg out.pfm\n", argv[0]); return 2; }
    int w, h, c;
    float *d = stbi_loadf(argv[1], &w, &h, &c, 4);
    if (!d) { fprintf(stderr, "stbi_loadf failed: %s\n", stbi_failure_reason()); return 1; }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 1;
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    for (int y = h - 1; y >= 0; --y)
        for (int x = 0; x < w; ++x) fwrite(d + ((size_t)y * w + x) * 4, sizeof(float), 3, f);
    fclose(f);
    stbi_image_free(d);
    return 0;
}

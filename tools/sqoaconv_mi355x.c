/*
 * sqoaconv_mi355x.c -- counterpart of the reference's sqoaconv.c (sqoaconv.c:38-100) on
 * libsqoa_mi355x.so.  The .sqoa / .qoi legs are the reference's own calls:
 * sqoa_read(path, &desc, 0) (sqoaconv.c:65) and sqoa_write(path, pixels, &desc) with
 * qoi_compat taken from the ".qoi" extension (sqoaconv.c:84-90).  The PNG / JPG legs need stb and
 * tiny_jpeg, which are not vendored; raw pixels go through netpbm PAM (P7) instead.
 *
 *   sqoaconv_mi355x input.pam output.sqoa     sqoaconv_mi355x input.sqoa output.qoi
 *   sqoaconv_mi355x input.qoi output.pam
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SQOA_IMPLEMENTATION
#include "seqoia.h"

#define ENDS_WITH(S, E) (strlen(S) >= sizeof(E) - 1 && strcmp(S + strlen(S) - (sizeof(E) - 1), E) == 0)

static void *pam_read(const char *path, int *w, int *h, int *ch) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    char line[256];
    int maxval = 0, ok = 0;
    *w = *h = *ch = 0;
    if (!fgets(line, sizeof line, f) || strncmp(line, "P7", 2)) { fclose(f); return NULL; }
    while (fgets(line, sizeof line, f)) {
        if (!strncmp(line, "ENDHDR", 6)) { ok = 1; break; }
        sscanf(line, "WIDTH %d", w); sscanf(line, "HEIGHT %d", h); sscanf(line, "DEPTH %d", ch); sscanf(line, "MAXVAL %d", &maxval);
    }
    if (!ok || *w <= 0 || *h <= 0 || *ch < 1 || *ch > 4 || maxval != 255) { fclose(f); return NULL; }
    size_t n = (size_t)*w * *h * *ch;
    void *px = malloc(n);
    if (px && fread(px, 1, n, f) != n) { free(px); px = NULL; }
    fclose(f);
    return px;
}

static int pam_write(const char *path, const void *px, int w, int h, int ch) {
    static const char *tupl[5] = {"", "GRAYSCALE", "GRAYSCALE_ALPHA", "RGB", "RGB_ALPHA"};
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    fprintf(f, "P7\nWIDTH %d\nHEIGHT %d\nDEPTH %d\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n", w, h, ch, tupl[ch]);
    size_t n = (size_t)w * h * ch;
    int ok = fwrite(px, 1, n, f) == n;
    fclose(f);
    return ok;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        puts("Usage: sqoaconv_mi355x <infile> <outfile>");
        puts("Examples:\n  sqoaconv_mi355x input.pam output.sqoa\n  sqoaconv_mi355x input.qoi output.pam\n  sqoaconv_mi355x input.sqoa output.qoi");
        return 1;
    }
    void *pixels = NULL;
    int w = 0, h = 0, channels = 0;
    if (ENDS_WITH(argv[1], ".pam")) pixels = pam_read(argv[1], &w, &h, &channels);
    else if (ENDS_WITH(argv[1], ".sqoa") || ENDS_WITH(argv[1], ".qoi")) {
        sqoa_desc desc;
        pixels = sqoa_read(argv[1], &desc, 0);
        channels = (desc.channels < 3 ? 1 : 3) + ((desc.channels & 1) ? 0 : 1);
        w = desc.width;
        h = desc.height;
    }
    if (!pixels) { printf("Couldn't load/decode %s\n", argv[1]); return 1; }
    int encoded = 0;
    if (ENDS_WITH(argv[2], ".pam")) encoded = pam_write(argv[2], pixels, w, h, channels);
    else if (ENDS_WITH(argv[2], ".sqoa") || ENDS_WITH(argv[2], ".qoi"))
        encoded = sqoa_write(argv[2], pixels, &(sqoa_desc){.width = w, .height = h, .channels = channels, .colorspace = SQOA_SRGB,
                                                            .qoi_compat = ENDS_WITH(argv[2], ".qoi")});
    if (!encoded) { printf("Couldn't write/encode %s\n", argv[2]); return 1; }
    free(pixels);
    return 0;
}

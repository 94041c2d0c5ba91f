/* A C caller of libsqoa_mi355x.so in the shape of the reference's fuzz harness
 * (sqoafuzz.c:20-31): every record is `int channels` (4 bytes, host order) followed by
 * a stream, handed to sqoa_decode exactly as LLVMFuzzerTestOneInput does.  Unlike the
 * harness it reports what came back, so that a test can compare it with the golden
 * vectors:  one line per record,  "<idx> NULL"  or
 * "<idx> <w> <h> <file channels> <colorspace> <qoi_compat> <bytes> <fnv1a64 hex>".
 *
 * Input file: records of  u32 length (little endian) + `length` bytes.
 * With -DWITH_FUZZ_ENTRY the record is ALSO passed to an external
 * LLVMFuzzerTestOneInput (the reference's own sqoafuzz.c, compiled unchanged and
 * linked in by tests/test_dropin.py), which must simply return. */
#define SQOA_IMPLEMENTATION
#include "seqoia.h"
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef WITH_FUZZ_ENTRY
int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size);
#endif

static int fuzz_shaped_call(const uint8_t *data, size_t size, int idx) {
    sqoa_desc desc;
    int channels;
    void *decoded;
    if (size < 4) {
        return 0;
    }
    memcpy(&channels, data, 4);
    memset(&desc, 0, sizeof desc);
    decoded = sqoa_decode((const void *)(data + 4), (int)(size - 4), &desc, channels);
    if (decoded == NULL) {
        printf("%d NULL\n", idx);
        return 0;
    }
    {
        int col = desc.channels < 3 ? 1 : 3;
        int oc = channels ? channels : col + (desc.channels % 2 == 0);
        size_t n = (size_t)desc.width * desc.height * (size_t)oc, i;
        uint64_t h = 0xcbf29ce484222325ull;
        for (i = 0; i < n; i++) {
            h = (h ^ ((const uint8_t *)decoded)[i]) * 0x100000001b3ull;
        }
        printf("%d %u %u %u %u %u %zu %016llx\n", idx, desc.width, desc.height, desc.channels, desc.colorspace,
               desc.qoi_compat, n, (unsigned long long)h);
    }
    free(decoded);
    return 0;
}

int main(int argc, char **argv) {
    FILE *f;
    int idx = 0;
    if (argc != 2 || !(f = fopen(argv[1], "rb"))) {
        return 2;
    }
    for (;; idx++) {
        uint8_t lenb[4], *buf;
        uint32_t len;
        if (fread(lenb, 1, 4, f) != 4) {
            break;
        }
        len = lenb[0] | lenb[1] << 8 | lenb[2] << 16 | (uint32_t)lenb[3] << 24;
        buf = malloc(len ? len : 1);
        if (!buf || fread(buf, 1, len, f) != len) {
            return 3;
        }
#ifdef WITH_FUZZ_ENTRY
        if (LLVMFuzzerTestOneInput(buf, len) != 0) {
            return 4;
        }
#endif
        fuzz_shaped_call(buf, len, idx);
        free(buf);
    }
    fclose(f);
    return 0;
}

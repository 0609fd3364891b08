/* Driver that exposes the reference's vendored xxHash 0.8.0 (compiled from
 * /root/reference/assembler/ext/include/xxh/xxhash.h where it lies) so tests can pin
 * oracle/bbk_oracle.c:orc_xxh3_64 and the HIP bucket function against it.
 * Test infrastructure only. */
#define XXH_INLINE_ALL
#include "xxh/xxhash.h"
#include <stdint.h>
#include <stddef.h>

uint64_t ref_xxh3_64_with_seed(const void *data, size_t len, uint64_t seed) {
    return XXH3_64bits_withSeed(data, len, seed);
}

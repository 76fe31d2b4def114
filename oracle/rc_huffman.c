/*
 * rc_huffman.c -- ORACLE (test infrastructure).  Restates the reference's Huffman primitives:
 *   Code.huffmanCodes(from:)        Sources/Common/CodingTree/Code.swift:15-39
 *   CodeLength ordering             Sources/Common/CodingTree/CodeLength.swift:13-22
 *   DecodingTree.init               Sources/Common/CodingTree/DecodingTree.swift:15-34
 *   DecodingTree.findNextSymbol()   Sources/Common/CodingTree/DecodingTree.swift:36-50
 *   Int.reversed(bits:)             Sources/Common/Extensions.swift:43-55
 *
 * Faithful to the reference's data structure on purpose (implicit heap of 2^(maxBits+1)-1 nodes,
 * one bit per step): the code sets are NOT validated, so an over-subscribed set overwrites earlier
 * leaves and a shorter code shadows longer ones -- the heap reproduces that
 * for free.
 */
#include "rc_common.h"

int rc_tree_build(rc_tree* t, const int* lengths, int n) {
    t->nodes = NULL;
    t->leaf_count = 0;
    /* maxBits = sortedLengths.last!.codeLength  (Code.swift:20) */
    int max_bits = lengths[0];
    for (int i = 1; i < n; i++) if (lengths[i] > max_bits) max_bits = lengths[i];
    if (max_bits < 0) max_bits = 0; /* every length <= 0: nothing is inserted, 1-node tree */
    /* max_bits > 20 is only reachable through bzip2's unchecked FINAL code length (BZip2.swift:185 runs before the deltas of
     * a symbol, so the last symbol's length is never range-checked).  The reference then builds the tree like any other:
     * (1 << (maxBits + 1)) - 1 Ints (DecodingTree.swift:19).  Followed up to maxBits 26 (1 GiB of Ints there, 512 MiB of
     * int32 here); beyond that the allocation is the outcome and the stream is classified trap-class. */
    if (max_bits > 26) return SWC_E_REF_TRAP;
    int64_t leaf_count = ((int64_t)1 << (max_bits + 1)) - 1; /* DecodingTree.swift:19 */
    int32_t* nodes = (int32_t*)malloc((size_t)leaf_count * sizeof(int32_t));
    if (!nodes) return SWC_E_REF_TRAP;
    memset(nodes, 0xFF, (size_t)leaf_count * sizeof(int32_t)); /* -1 */

    /* iterate in (codeLength, symbol) order, skipping codeLength <= 0 (Code.swift:26) */
    int loop_bits = -1;
    uint64_t symbol = (uint64_t)-1; /* Swift Int, wrapping shifts */
    for (int bits = 1; bits <= max_bits; bits++) {
        for (int s = 0; s < n; s++) {
            if (lengths[s] != bits) continue;
            symbol += 1;
            if (bits != loop_bits) {
                int sh = bits - loop_bits;
                symbol = sh >= 64 ? 0 : symbol << sh;
                loop_bits = bits;
            }
            /* code = symbol.reversed(bits:) then walked LSB-first == low `bits` bits of symbol
             * walked MSB-first (DecodingTree.swift:24-31). */
            int64_t index = 0;
            for (int k = bits - 1; k >= 0; k--) {
                int bit = (int)((symbol >> k) & 1);
                index = 2 * index + 1 + bit;
            }
            nodes[index] = s; /* later codes overwrite earlier ones (App. A1) */
        }
    }
    t->nodes = nodes;
    t->leaf_count = leaf_count;
    return SWC_OK;
}

void rc_tree_free(rc_tree* t) {
    free(t->nodes);
    t->nodes = NULL;
    t->leaf_count = 0;
}

int rc_tree_next(const rc_tree* t, rc_bits* r) {
    int64_t bits_left = rc_bits_left(r);
    int64_t index = 0;
    while (bits_left > 0) {
        int bit = rc_bit(r);
        index = 2 * index + 1 + bit;
        bits_left--;
        if (index >= t->leaf_count) return -1;
        if (t->nodes[index] > -1) return t->nodes[index];
    }
    return -1;
}

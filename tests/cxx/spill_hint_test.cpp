// Host-only check of the spill-feedback bookkeeping (ed_workspace.h: SpillHint) -- no HIP call is made:
// the "pinned slot" is a local variable.  Built and run by tests/test_library.py.
#include <cstdio>

#include "ed_workspace.h"

#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            std::printf("FAILED line %d: %s\n", __LINE__, #c);            \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main()
{
    using ed::SpillHint;
    unsigned long long slot = 0;
    SpillHint h;
    h.host = &slot;
    const unsigned long long A = 0x1111, B = 0x2222;
    CHECK(h.fraction(A) == 0.f && !h.known(A));
    const unsigned s1 = h.begin_call(A, 1000);
    CHECK(s1 != 0);
    h.absorb();                                    // nothing reported yet
    CHECK(!h.known(A));
    const unsigned s2 = h.begin_call(B, 500);
    CHECK(s2 != s1 && s2 != 0);
    slot = ((unsigned long long)s1 << 32) | 250;   // the device reports call s1: 250 tiles beyond the box
    h.absorb();
    CHECK(h.known(A) && !h.known(B));
    CHECK(h.fraction(A) > 0.249f && h.fraction(A) < 0.251f);
    h.absorb();                                    // the same report again: consumed, nothing changes
    CHECK(h.fraction(A) > 0.249f && h.fraction(A) < 0.251f);
    slot = ((unsigned long long)s2 << 32) | 0;     // call s2: no tile beyond the box
    h.absorb();
    CHECK(h.known(B) && h.fraction(B) == 0.f);
    slot = ((unsigned long long)0xdeadbeefu << 32) | 77;      // garbage (a fresh workspace): unknown sequence number
    h.absorb();
    CHECK(h.fraction(A) > 0.249f && h.fraction(B) == 0.f);
    // a later call of geometry A replaces its entry
    const unsigned s3 = h.begin_call(A, 1000);
    slot = ((unsigned long long)s3 << 32) | 10;
    h.absorb();
    CHECK(h.fraction(A) > 0.009f && h.fraction(A) < 0.011f);
    // more geometries than table entries: the oldest is forgotten, the newest is there
    for (unsigned k = 0; k < 12; ++k) {
        const unsigned s = h.begin_call(0x9000 + k, 100);
        slot = ((unsigned long long)s << 32) | (k + 1);
        h.absorb();
    }
    CHECK(h.known(0x9000 + 11) && h.fraction(0x9000 + 11) > 0.119f);
    CHECK(!h.known(A));
    // reports older than the ring (8 calls) are ignored
    const unsigned old = h.begin_call(A, 1000);
    for (unsigned k = 0; k < 8; ++k)
        h.begin_call(0x7000 + k, 10);
    slot = ((unsigned long long)old << 32) | 999;
    h.absorb();
    CHECK(!h.known(A));
    // the sequence number never becomes 0 (0 = "no report")
    h.seq = 0xffffffffu;
    CHECK(h.begin_call(A, 1) != 0);
    std::printf("ok\n");
    return 0;
}

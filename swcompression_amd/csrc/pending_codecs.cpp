// pending_codecs.cpp -- entry points whose kernels are not built yet in this revision report
// SWC_E_DEVICE ("no device path") instead of silently decoding on the CPU.  Each one moves to its own
// framing_<codec>.cpp when its kernel lands; this file must be empty by the end of the round.
#include "framing.h"
using namespace swc;
extern "C" {
#define PENDING_OUT(out, out_len) do { give_empty(out, out_len); return SWC_E_DEVICE; } while (0)
int swc_unarchive_many(int, const uint8_t* const*, const size_t*, size_t, uint8_t**, size_t*, int32_t*) { return SWC_E_DEVICE; }
}

// ABI version + error strings.
#include "common.hpp"

extern "C" int mmssl_abi_version(void) { return MMSSL_ABI_VERSION; }

extern "C" const char* mmssl_strerror(int code) {
  switch (code) {
    case 0: return "success";
    case MMSSL_E_BADARG: return "mmssl: bad argument (null/misaligned pointer, negative size or malformed CSR)";
    case MMSSL_E_UNSUPP: return "mmssl: unsupported shape (feature width / sizes outside the documented set)";
    case MMSSL_E_WORKSPACE: return "mmssl: workspace missing or too small";
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "mmssl: unknown error";
}

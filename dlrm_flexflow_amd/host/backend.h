// backend.h -- dlopen loader for a library exporting the kernel C-ABI (include/ff_hip.h).
// The product default is libffhip.so (hand-written gfx950 HIP); loading fails loudly if it is
// missing or incomplete -- there is no CPU fallback in this layer.
#pragma once
#include <string>
#include "../../include/ff_hip.h"

struct KernelApi {
#define FFH_DECL(name) decltype(&::name) name;
  FFH_API_LIST(FFH_DECL)
#undef FFH_DECL
  void* handle;
  std::string path;
  bool overridden = false;      // chosen by --backend or $FFH_BACKEND_LIB rather than the product default: the driver says so on its THROUGHPUT line
};

// Loads `path` (or, when empty, $FFH_BACKEND_LIB, else csrc/libffhip.so next to this library).
// Aborts with a message naming the missing file/symbol.
const KernelApi* load_kernel_api(const std::string& path);
std::string default_backend_path();

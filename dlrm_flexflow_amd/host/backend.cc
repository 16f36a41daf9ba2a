#include "backend.h"

#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

// FFH_BACKEND_LIB points the driver at another library exporting include/ff_hip.h (the CPU oracle in tests, an A/B build).  It is
// the one environment variable this layer reads, so it is never silent: load_kernel_api() says on stderr which library an
// override selected, and the driver names the kernel library on its THROUGHPUT line whenever it is not the product's own
// (KernelApi::overridden).
static bool g_env_override = false;
std::string default_backend_path() {
  if (const char* e = getenv("FFH_BACKEND_LIB")) { g_env_override = true; return std::string(e); }
  Dl_info info;
  std::string dir = ".";
  if (dladdr((void*)&default_backend_path, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.rfind('/');
    if (k != std::string::npos) dir = p.substr(0, k);
  }
  return dir + "/../csrc/libffhip.so";
}

const KernelApi* load_kernel_api(const std::string& path_in) {
  static std::map<std::string, KernelApi*>& cache = *new std::map<std::string, KernelApi*>();   // one table per library for the life of the process
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const std::string path = path_in.empty() ? default_backend_path() : path_in;
  auto it = cache.find(path);
  if (it != cache.end()) return it->second;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "FATAL: cannot load kernel library %s: %s\n"
                    "       build it with: python -c 'import __graft_entry__ as g; g.build()'\n", path.c_str(), dlerror());
    abort();
  }
  KernelApi* api = new KernelApi();
  api->handle = h;
  api->path = path;
  api->overridden = !path_in.empty() || g_env_override;
#define FFH_LOAD(name)                                                        \
  api->name = reinterpret_cast<decltype(api->name)>(dlsym(h, #name));         \
  if (!api->name) {                                                           \
    fprintf(stderr, "FATAL: %s does not export %s\n", path.c_str(), #name);   \
    abort();                                                                  \
  }
  FFH_API_LIST(FFH_LOAD)
#undef FFH_LOAD
  if (api->ffh_abi_version() != FFH_ABI_VERSION) {
    fprintf(stderr, "FATAL: %s has ABI version %d, expected %d\n", path.c_str(), api->ffh_abi_version(), FFH_ABI_VERSION);
    abort();
  }
  if (path_in.empty() && g_env_override)
    fprintf(stderr, "[DLRM] FFH_BACKEND_LIB: kernel library %s (%s)\n", path.c_str(), api->ffh_backend_name());
  cache[path] = api;
  return api;
}

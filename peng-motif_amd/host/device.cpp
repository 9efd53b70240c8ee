#include "device.h"

#include <cstdlib>
#include <iostream>

#include "Global.h"

namespace pengk_host {

static pengk_ctx* g_ctx = nullptr;

void check(int rc, const char* what) {
  if (rc == PENGK_OK) return;
  std::cerr << "Error: " << what << " failed: " << pengk_error_name(rc) << ": " << pengk_last_error() << std::endl;
  exit(1);
}

pengk_ctx* context() {
  if (!g_ctx) check(pengk_create(Global::device, &g_ctx), "pengk_create");
  return g_ctx;
}

void shutdown() {
  if (g_ctx) pengk_destroy(g_ctx);
  g_ctx = nullptr;
}

}  // namespace pengk_host

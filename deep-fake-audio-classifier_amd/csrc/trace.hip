// trace.hip -- see trace.h
#include "trace.h"

#include <dlfcn.h>
#include <stdlib.h>

namespace dfa {

namespace {
typedef int (*push_fn)(const char*);
typedef int (*pop_fn)();
struct Roctx {
  push_fn push = nullptr;
  pop_fn pop = nullptr;
  Roctx() {
    const char* e = getenv("DFA_ROCTX");
    if (!e || e[0] == '0' || e[0] == 0) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (push_fn)dlsym(h, "roctxRangePushA");
    pop = (pop_fn)dlsym(h, "roctxRangePop");
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
const Roctx& roctx() {
  static const Roctx r;
  return r;
}
}  // namespace

TraceRange::TraceRange(const char* name) : on_(roctx().push != nullptr) {
  if (on_) roctx().push(name);
}
TraceRange::~TraceRange() {
  if (on_) roctx().pop();
}

}  // namespace dfa

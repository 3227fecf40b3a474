// trace.h -- optional roctx ranges around the C-ABI entry points (the reference has no tracing; SURVEY.md section 5 lists it as
// an aux subsystem).  With DFA_ROCTX=1 in the environment every forward / backward / optimiser call pushes a named range, so
// `rocprofv3 --marker-trace --kernel-trace` groups the kernels by ABI call.  The marker library is looked up at run time
// (librocprofiler-sdk-roctx.so, then libroctx64.so): no link-time dependency, and without the variable the cost is one branch.
#pragma once

namespace dfa {

struct TraceRange {
  explicit TraceRange(const char* name);
  ~TraceRange();
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;

 private:
  bool on_;
};

}  // namespace dfa

// abort_trace.c -- LD_PRELOAD aid: the C backtrace of a SIGABRT / SIGSEGV (faulthandler shows the Python frames only).
//   gcc -shared -fPIC -o build/abort_trace.so tools/abort_trace.c;  LD_PRELOAD=$PWD/build/abort_trace.so python -m pytest ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_fatal(int sig) {
  void *frames[64];
  const char msg[] = "\n== abort_trace: fatal signal, C backtrace ==\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fatal;
  sigaction(SIGABRT, &sa, NULL);
  sigaction(SIGSEGV, &sa, NULL);
}

/* LD_PRELOAD diagnostic: print the C backtrace of the thread that calls abort() (a GPU-runtime thread has no Python frame for
 * faulthandler to show).  gcc -shared -fPIC -o abort_trace.so tools/abort_trace.c ; run pytest with -p no:faulthandler. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>

static void on_abort(int sig) {
  void* bt[96];
  const char msg[] = "\n==== abort_trace: backtrace of the aborting thread ====\n";
  write(2, msg, sizeof(msg) - 1);
  int n = backtrace(bt, 96);
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, 0);
}

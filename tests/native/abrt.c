/* abrt.c -- test infrastructure: a SIGABRT handler that writes the C call stack to stderr before the previous handler
 * (pytest's faulthandler: the Python stack) runs.  One of ~12 full `pytest -m gpu` runs of round 3 died with a silent abort()
 * inside the HIP runtime; should it happen again, the log says where.  Loaded by tests/conftest.py when present. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction previous;
static int out_fd = 2;

static void on_abort(int sig, siginfo_t* info, void* ctx) {
	static const char head[] = "\n==== SIGABRT: C call stack (tests/native/abrt.c) ====\n";
	void* frames[96];
	int n;
	if (write(out_fd, head, sizeof(head) - 1) < 0) { /* nothing to do about it */ }
	n = backtrace(frames, 96);
	backtrace_symbols_fd(frames, n, out_fd);
	if (previous.sa_flags & SA_SIGINFO) {
		if (previous.sa_sigaction) previous.sa_sigaction(sig, info, ctx);
	} else if (previous.sa_handler != SIG_DFL && previous.sa_handler != SIG_IGN && previous.sa_handler) {
		previous.sa_handler(sig);
	}
	signal(SIGABRT, SIG_DFL);  /* abort() re-raises with the default disposition after the handler returns anyway */
}

/* fd: where to write (pytest captures fd 2 while a test runs; its faulthandler plug-in keeps a duplicate of the real stderr) */
int abrt_install(int fd) {
	if (fd >= 0) out_fd = fd;
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_abort;
	sa.sa_flags = SA_SIGINFO;
	sigemptyset(&sa.sa_mask);
	return sigaction(SIGABRT, &sa, &previous);
}

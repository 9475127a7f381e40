/* abrt.c -- test infrastructure: a handler for SIGABRT / SIGSEGV / SIGBUS that writes the name of the test that was running and the C call
 * stack to stderr before the previous handler (pytest's faulthandler: the Python stacks) runs.  One of ~12 full `pytest -m gpu` runs of
 * round 3 died with a silent abort() inside the HIP runtime, one of 17 in round 6 with a fatal signal whose log was cut to the Python frames
 * at the bottom of the main thread: should it happen again, the log says which test and where.  Loaded by tests/conftest.py when present. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction previous[3];
static const int signals[3] = {SIGABRT, SIGSEGV, SIGBUS};
static int out_fd = 2;
static char current_test[512] = "(no test has started)";
static char marker_path[512] = "";

static void put(int fd, const char* s) {
	if (write(fd, s, strlen(s)) < 0) { /* nothing to do about it */ }
}

static void on_fatal(int sig, siginfo_t* info, void* ctx) {
	void* frames[96];
	int n, k = 0;
	while (k < 3 && signals[k] != sig) k++;
	put(out_fd, "\n==== fatal signal ");
	put(out_fd, sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : "SIGBUS");
	put(out_fd, " while running: ");
	put(out_fd, current_test);
	put(out_fd, "\n==== C call stack of the faulting thread (tests/native/abrt.c) ====\n");
	n = backtrace(frames, 96);
	backtrace_symbols_fd(frames, n, out_fd);
	if (marker_path[0]) {  /* the same two facts in a file of their own: a log that is cut to its tail loses the lines above */
		int fd = open(marker_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (fd >= 0) {
			put(fd, sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : "SIGBUS");
			put(fd, " while running: ");
			put(fd, current_test);
			put(fd, "\n");
			backtrace_symbols_fd(frames, n, fd);
			close(fd);
		}
	}
	if (k < 3) {
		const struct sigaction* p = &previous[k];
		if (p->sa_flags & SA_SIGINFO) {
			if (p->sa_sigaction) p->sa_sigaction(sig, info, ctx);
		} else if (p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN && p->sa_handler) {
			p->sa_handler(sig);
		}
	}
	signal(sig, SIG_DFL);  /* abort() re-raises with the default disposition after the handler returns; a fault re-executes and dies */
}

/* fd: where to write (pytest captures fd 2 while a test runs; its faulthandler plug-in keeps a duplicate of the real stderr) */
int abrt_install(int fd) {
	int rc = 0, i;
	if (fd >= 0) out_fd = fd;
	for (i = 0; i < 3; i++) {
		struct sigaction sa;
		memset(&sa, 0, sizeof(sa));
		sa.sa_sigaction = on_fatal;
		sa.sa_flags = SA_SIGINFO;
		sigemptyset(&sa.sa_mask);
		rc |= sigaction(signals[i], &sa, &previous[i]);
	}
	return rc;
}

/* the test about to run (tests/conftest.py pytest_runtest_logstart) */
void abrt_set_test(const char* nodeid) {
	if (!nodeid) return;
	strncpy(current_test, nodeid, sizeof(current_test) - 1);
	current_test[sizeof(current_test) - 1] = 0;
}

/* a file that receives the test name and the C stack as well ("" = none) */
void abrt_set_marker(const char* path) {
	if (!path) { marker_path[0] = 0; return; }
	strncpy(marker_path, path, sizeof(marker_path) - 1);
	marker_path[sizeof(marker_path) - 1] = 0;
}

/* pcmp.c -- test infrastructure: parallel memcmp / memcpy of large host buffers on a small persistent pool of threads that
 * SLEEP between jobs (the full-size streaming test compares every delivered 256 MiB buffer while the next one is already on
 * its way; one thread cannot keep up with PCIe Gen5, and spinning helpers would eat the container's CPU quota).
 *   gcc -O2 -pthread -shared -fPIC pcmp.c -o libpcmp.so */
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define MAX_THREADS 64
#define PIECE ((size_t)1 << 20)

typedef struct {
	pthread_mutex_t m;
	pthread_cond_t wake, done;
	pthread_t th[MAX_THREADS];
	int nthreads, started;
	unsigned long gen;
	int pending;
	/* job */
	int op; /* 0 compare, 1 copy */
	const unsigned char* a;
	unsigned char* b;
	size_t n;
	long next_piece, pieces, bad;
} pool_t;

static pool_t g = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER};

static void work(pool_t* p) {
	for (;;) {
		pthread_mutex_lock(&p->m);
		const long i = p->next_piece < p->pieces ? p->next_piece++ : -1;
		pthread_mutex_unlock(&p->m);
		if (i < 0) return;
		const size_t lo = (size_t)i * PIECE, len = lo + PIECE <= p->n ? PIECE : p->n - lo;
		if (p->op == 1) {
			memcpy(p->b + lo, p->a + lo, len);
		} else if (memcmp(p->a + lo, p->b + lo, len) != 0) {
			pthread_mutex_lock(&p->m);
			p->bad++;
			pthread_mutex_unlock(&p->m);
		}
	}
}

static void* helper(void* arg) {
	pool_t* p = (pool_t*)arg;
	unsigned long seen = 0;
	for (;;) {
		pthread_mutex_lock(&p->m);
		while (p->gen == seen) pthread_cond_wait(&p->wake, &p->m);
		seen = p->gen;
		pthread_mutex_unlock(&p->m);
		work(p);
		pthread_mutex_lock(&p->m);
		if (--p->pending == 0) pthread_cond_signal(&p->done);
		pthread_mutex_unlock(&p->m);
	}
	return NULL;
}

/* one job at a time (callers serialise); the calling thread works too */
static long run(int op, const unsigned char* a, unsigned char* b, size_t n, int threads) {
	pool_t* p = &g;
	if (threads > MAX_THREADS) threads = MAX_THREADS;
	if (threads < 1) threads = 1;
	pthread_mutex_lock(&p->m);
	while (p->started < threads - 1) {
		pthread_create(&p->th[p->started], NULL, helper, p);
		p->started++;
	}
	p->op = op; p->a = a; p->b = b; p->n = n;
	p->pieces = (long)((n + PIECE - 1) / PIECE); p->next_piece = 0; p->bad = 0;
	p->pending = p->started;
	p->gen++;
	pthread_cond_broadcast(&p->wake);
	pthread_mutex_unlock(&p->m);
	work(p);
	pthread_mutex_lock(&p->m);
	while (p->pending != 0) pthread_cond_wait(&p->done, &p->m);
	const long bad = p->bad;
	pthread_mutex_unlock(&p->m);
	return bad;
}

static pthread_mutex_t job = PTHREAD_MUTEX_INITIALIZER;

/* number of 1 MiB pieces that differ (0 = identical) */
long pcmp(const unsigned char* a, const unsigned char* b, size_t n, int threads) {
	pthread_mutex_lock(&job);
	const long r = run(0, a, (unsigned char*)b, n, threads);
	pthread_mutex_unlock(&job);
	return r;
}

void pcopy(unsigned char* dst, const unsigned char* src, size_t n, int threads) {
	pthread_mutex_lock(&job);
	run(1, src, dst, n, threads);
	pthread_mutex_unlock(&job);
}

/* oracle/noma_params_main.c — TEST INFRASTRUCTURE (oracle).  Not part of the shipped library.
 *
 * NOMA.c has no command line: its parameters are initialised file-scope variables (NOMA.c:41-57).  This
 * driver is linked with the reference's NOMA.c compiled AS IT LIES under /root/reference with
 * -Dmain=noma_reference_main (nothing copied, nothing edited): it sets those variables from argv, makes
 * stdout line-buffered (the fuzz harness cuts runs off by wall clock) and calls the reference's own main.
 * Output of `make -C oracle ref`: oracle/_ref/NOMA_params.  Used by tests/golden/fuzz_reference_noma.py only.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern int nPreamble, backoffIndicator, nGrantUL, maxRarWindow, maxMsg1ReTx, accessTime; /* NOMA.c:41-47 */
extern float cellRadius;                                                                /* NOMA.c:56 */
int noma_reference_main(int argc, char **argv);

int main(int argc, char **argv) {
    for (int i = 1; i + 1 < argc; i += 2) {
        const char *k = argv[i], *v = argv[i + 1];
        if (!strcmp(k, "-p")) nPreamble = atoi(v);
        else if (!strcmp(k, "-b")) backoffIndicator = atoi(v);
        else if (!strcmp(k, "-g")) nGrantUL = atoi(v);
        else if (!strcmp(k, "-rw")) maxRarWindow = atoi(v);
        else if (!strcmp(k, "-m")) maxMsg1ReTx = atoi(v);
        else if (!strcmp(k, "-s")) accessTime = atoi(v);
        else if (!strcmp(k, "-c")) cellRadius = (float)atof(v);
        else { fprintf(stderr, "unknown option %s\n", k); return 2; }
    }
    setvbuf(stdout, NULL, _IOLBF, 0);
    char *av[2] = { argv[0], NULL };
    return noma_reference_main(1, av);
}

#!/bin/sh
# oracle/check_ref_lines.sh REF_DIR — TEST INFRASTRUCTURE.  The sed recipes of oracle/Makefile address reference lines by NUMBER; this
# refuses (exit 1) to let them run against a revision of the reference in which those lines do not hold the text they were written for.
# (The reference tree has no .git; the recipes were written against the files whose SHA-256 oracle/Makefile lists.)
REF="${1:?reference directory}"
rc=0
expect() { # file line text
    if ! sed -n "${2}p" "$REF/$1" | grep -qF -- "$3"; then
        echo "oracle/check_ref_lines.sh: line $2 of $1 does not contain '$3' — the sed recipes were written against another revision of the reference" >&2
        rc=1
    fi
}
expect RandomAccessWithNOMA.c 312 '// preambleCollision(UE + i, UE, nUE, checkPreambleNumber, nPreamble, time, backoffIndicator, maxRarWindow, sectorGrants, nGrantUL);'
expect RandomAccessWithNOMA.c 313 'preambleCollision(UE + i, UE, nUE, checkPreambleNumber, nPreamble, time, backoffIndicator, maxRarWindow, &grantCheck, nGrantUL);'
expect RandomAccessWithNOMA.c 626 '// int sector = user->sector;'
expect RandomAccessWithNOMA.c 637 '// }'
expect RandomAccessWithNOMA.c 639 '*grantCheck = *grantCheck + 1;'
expect RandomAccessWithNOMA.c 648 '}'
expect NOMA.c 688 '// preambleCollisionDetection(UEs, activeCheck, time, &grantCheck);'
expect NOMA.c 689 'preambleSectorCollisionDetection(UEs, activeCheck, time, sectorGrants);'
expect RandomAccessSimulatorBeta.c 71 'for (int n = 10000; n <= 100000; n += 10000){'
exit $rc

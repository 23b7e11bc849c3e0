/* oracle/oracle_cli.c -- TEST INFRASTRUCTURE: command-line front end of the CPU
 * restatement, same three modes as oracle/ref_driver.c so that tests can diff
 * the two (and the HIP engine) on identical inputs. */
#include <stdio.h>
#include <string.h>

#include "pip_oracle.h"

int main(int argc, char **argv) {
  if (argc >= 3 && !strcmp(argv[1], "dat")) {
    int simplify = 0, deepest = 0, a = 2, rc;
    FILE *in;
    if (!strcmp(argv[a], "-d")) {
      deepest = 1;
      a++;
    }
    if (!strcmp(argv[a], "-z")) {
      simplify = 1;
      a++;
    }
    in = fopen(argv[a], "r");
    if (!in) {
      fprintf(stderr, "%s unaccessible\n", argv[a]);
      return 1;
    }
    rc = ora_run_dat(in, stdout, simplify, deepest);
    fclose(in);
    return rc;
  }
  if (argc >= 2 && !strcmp(argv[1], "pip")) return ora_run_pip(stdin, stdout);
  if (argc >= 4 && !strcmp(argv[1], "batch")) return ora_run_batch(argv[2], argv[3]);
  fprintf(stderr, "usage: oraclepip dat [-d] [-z] in.dat | pip < in.pip | batch in.bin out.bin\n");
  return 64;
}

[PIP2-like future input] Please enter:
- the context matrix,
1 3
   1   1  -1
- the bignum column (start at 0, -1 if no bignum),
-1
- the constraint matrix.
2 4
   1   1   0   1
   1  -1   1   0

(list
 #[ 0 0]
)

[PIP2-like future input] Please enter:
- the context matrix,
0 3
- the bignum column (start at 0, -1 if no bignum),
-1
- the constraint matrix.
5 6
   1   1  -1   2   0   0
   1   0   1   1   4  20
   1   0  -1  -1   0   0
   1   0   1  -1   2  10
   1   0  -1   1   2  10

(if #[ 1 5]
 (list
  #[ -3 -15]
  #[ -1 -5]
  #[ 1 5]
 )
 ()
)

[PIP2-like future input] Please enter:
- the context matrix,
0 2
- the bignum column (start at 0, -1 if no bignum),
-1
- the constraint matrix.
4 4
 1 1 0 0
 1 0 1 0
 1 1 -3 12
 1 -2 1 3

(list
 #[ 0]
 #[ 0]
)
 (list
  #[ 0]
  #[ 0]
  #[ 0]
  #[ 0]
 )

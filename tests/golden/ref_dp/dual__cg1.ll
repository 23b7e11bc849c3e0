[PIP2-like future input] Please enter:
- the context matrix,
2 4
 1 1 0 -1
 1 -1 1 0
- the bignum column (start at 0, -1 if no bignum),
-1
- the constraint matrix.
8 7
 1 0 1 0 -1 0 0
 1 0 -1 0 1 0 0
 1 1 0 0 0 -1 0
 1 -1 0 0 0 1 0
 1 0 1 0 0 0 -1
 1 0 -1 0 0 1 0
 1 0 -1 1 0 0 -1
 1 0 0 -1 0 1 0

(if #[ -1 1 -1]
 (list
  #[ 0 1 0]
  #[ 1 0 0]
  #[ 1 0 1]
 )
  (list
   #[ 0]
   #[ 0]
   #[ 0]
   #[ 0]
   #[ 0]
   #[ 0]
   #[ 0]
   #[ 0]
  )
 ()
)

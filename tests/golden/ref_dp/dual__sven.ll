[PIP2-like future input] Please enter:
- the context matrix,
0 3
- the bignum column (start at 0, -1 if no bignum),
-1
- the constraint matrix.
2 3
 1 1 -4
 1 -1 10

(if #[ 1 -4]
 (if #[ -1 10]
  (list
  )
   (list
    #[ 0]
    #[ 0]
   )
  ()
 )
 ()
)

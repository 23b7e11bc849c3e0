( 
 ( Solving MIN(i-2.j) under the following constraints:
   Unknowns may be negative.
   Order:
   f' i' j' constant G P n'
  )
(if #[ 0 -1 1 5]
(list #[ 1 3 -3 -15]
#[ 1 1 -1 -5]
#[ 1 -1 1 5]
)
()
)
)

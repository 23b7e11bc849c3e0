( (Lower bound on j after loop inversion
      (unknowns j i)
      (parameters k m n))
(if #[ -1 1 0 0]
(list #[ 0 0 0 0]
#[ 1 0 0 0]
)
(list #[ 1 -1 0 0]
#[ 0 1 0 0]
)
)
)

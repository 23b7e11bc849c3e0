((test du papier RAIRO)
(if #[ -1 2 1 0]
(if #[ 1 -2 0 0]
(list #[ 0 0 0 0]
#[ -1 2 1 0]
)
(list #[ -1/2 1 0 0]
#[ 0 0 1 0]
)
)
()
)
)

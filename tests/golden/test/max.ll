((a maximization problem)
(if #[ -1 3]
(list #[ 0 0]
#[ 0 0]
)
(if #[ -1 5]
(newparm 1 (div #[ 1 1]
 2)
)
(list #[ 1 -1 -1]
#[ 0 0 0]
)
(list #[ 1 -4]
#[ 1 -5]
)
)
)
)

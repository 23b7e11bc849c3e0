((a maximization problem)
(list #[ 1 -4]
#[ 1 -5]
)
)

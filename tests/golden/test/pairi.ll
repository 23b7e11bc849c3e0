((les nombres pairs)
(newparm 1 (div #[ 1 0]
 2)
)
(if #[ -1 2 0]
(list #[ 1 -1 0]
)
()
)
)

((linearisation d'un tableau 100 100)
(if #[ -1 99]
(list #[ 0 0]
)
(newparm 1 (div #[ 99 99]
 100)
)
(list #[ 1 -1 0]
)
)
)

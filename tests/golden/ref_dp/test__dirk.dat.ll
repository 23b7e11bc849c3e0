(((i j 1)(m n))
(if #[ -1 0 1]
(if #[ 0 -1 0]
()
(if #[ 0 -1 5]
(if #[ -3 -1 5]
(if #[ 0 -1 1]
(newparm 2 (div #[ 0 1 1]
 3)
)
(newparm 3 (div #[ 0 1 5 3]
 6)
)
(newparm 4 (div #[ 0 2 0 1 0]
 3)
)
()
(if #[ 0 -1 3]
(newparm 2 (div #[ 0 2 1]
 3)
)
()
()
)
)
(if #[ 0 -1 3]
(newparm 2 (div #[ 0 2 1]
 3)
)
()
()
)
)
()
)
)
()
)
)

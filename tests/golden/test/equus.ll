((equation impossible)
(if #[ -1 0]
()
()
)
)

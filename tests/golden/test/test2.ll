((variables : a b c d ...., pas de parametres)
(list #[ 2]
#[ 1/2]
)
)

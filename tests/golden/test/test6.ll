((variables : a b c d ...., pas de parametres)
(list #[ 2]
#[ 1/2]
#[ 4/3]
#[ 9/2]
#[ 96/5]
#[ 100]
)
)

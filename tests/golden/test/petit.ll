((( a b c d e f g h i j k l m n o p q r s t u v w x y z 1)())
(list #[ 2]
#[ 1]
#[ 1]
)
)
